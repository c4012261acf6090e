// Compact ("fast") tier of the level-2 automaton: tables and launch parameters shared by the host
// (l2_fast_tables.cpp, capi_l2.cpp) and the kernel (l2_fast_kernel.hip).
//
// The general kernel (l2_kernel.hip) keeps every record of a document's StateMachine in an HBM arena
// and handles every program shape the reference accepts.  Rule sets whose programs are all FLAT --
// terms only, no sub-expression feeding another rule -- need far less state per rule instance: the
// events a rule can take are input lexems, so a captured item or the start of a match is an index into
// the document's lexem array, nothing is reference counted, nothing is joined.  For such rule sets
// (the reference's own benchmark shape, tests/randomTokenPatternMatch, and the token-rule stage of the
// pipeline) the whole hot state of a document -- rule words, trigger buckets, expiry lists, stop-word
// log -- fits a few KB and lives in LDS; HBM sees the lexems once, the install lines (one cache-resident 64-byte
// line per installed program) and the results.  A rule instance has a record in HBM only when it captured
// something the key lexem and its install line do not tell (round 3; a 32-byte record per instance before).
#ifndef SPA_L2_FAST_H
#define SPA_L2_FAST_H
#include <stdint.h>
#include "l2_tables.h"

namespace spa {

// ---- compiled tables of the fast tier (read only, HBM/L2) ----
// One record per (key event, program keyed by it), in the reference's visiting order of the key
// event's program list (src/ruleMatcherAutomaton.cpp:1137-1157): everything installProgram
// (cpp:1168-1270) needs, in one 64-byte line -- no second or third dependent table read.
struct FastKeyInst			// 64 B
{
	uint32_t resultHandle;		// 0: a match emits nothing
	uint32_t formatHandle;
	uint32_t pastEvent;		// original key event of an alternative-keyed program (cpp:1253-1258), 0 = none
	uint32_t meta;			// initvalue(4) | initcount(5)<<4 | range(6)<<9 | ntrig(2)<<15 | pastStopIdx(8)<<17
	struct { uint32_t event; uint32_t info; } trig[ 3];	// installation order (= last expression argument first)
					// info: sigval(4) | sigtype(3)<<4 | install(1)<<7 | key(1)<<8 | bucket(4)<<9 | hasVar(1)<<13 | delStopIdx(8)<<14 | variable(8)<<24
	// ---- STATIC INSTALL (round 3).  For a program that is not alternative-keyed, everything installProgram does
	// except the absolute positions is decided by the (key event, program) pair alone: the slot after the key trigger(s)
	// have fired, whether a result / a dispose entry is due at once, and -- because the 64 programs of a batch are
	// the key list entries [kiBegin + 64 b, ..) whatever the document -- the RANK of every record the batch appends
	// (bucket entries per bucket, expiry row entries per range, results, dispose entries).  The host simulates the key
	// fires (fireLocal below, the very function the kernel runs for the dynamic cases) and counts the ranks, so a
	// static batch needs no simulation, no prefix sum and no ballot on the device.
	uint32_t hw0;			// rule word after the key fires: value | count | done | active | trigger mask | nItems | hasStart (end bits excluded)
	uint32_t ranksA;		// expRank(8) | expClose(8): members of my expiry group if I am its last lane, else 0 | resRank(8) | dispRank(8)
	uint32_t ranksB;		// bRank0(8) | bRank1(8) | bRank2(8): rank of my template j among the batch's entries of its bucket | bLast(3)<<24: template j is its bucket's last
	uint32_t totals;		// of the batch: installed triggers(8) | key fires(8) | results(8) | dispose entries(8)
	uint32_t items;			// var0 | var1<<8 | var2<<16 (capture order) | items of the batch's immediate results (8)<<24
	uint32_t flags;			// FKF_*
};
enum {
	FKI_VALUE_MASK=0xFu, FKI_COUNT_SHIFT=4, FKI_COUNT_MASK=0x1Fu, FKI_RANGE_SHIFT=9, FKI_RANGE_MASK=0x3Fu,
	FKI_NTRIG_SHIFT=15, FKI_NTRIG_MASK=0x3u, FKI_PASTSTOP_SHIFT=17, FKI_PASTSTOP_MASK=0xFFu,
	FTI_SIGVAL_MASK=0xFu, FTI_SIGTYPE_SHIFT=4, FTI_SIGTYPE_MASK=0x7u, FTI_INSTALL=1u<<7, FTI_KEY=1u<<8,
	FTI_BUCKET_SHIFT=9, FTI_HASVAR=1u<<13, FTI_DELSTOP_SHIFT=14, FTI_DELSTOP_MASK=0xFFu, FTI_VAR_SHIFT=24,
	FKF_BATCH_STATIC=1u,		// every line of my 64-lane batch is static: the batch takes the static path
	FKF_END_SET=2u,			// a key fire took the event: end_ordpos = position + 1
	FKF_START_SET=4u,		// ... and set the start of the match to the key lexem
	FKF_RESULT_NOW=8u,		// the key fire completes the rule and the pattern is visible: a result at once
	FKF_DISPOSE_NOW=16u		// finished or deleted by its own key event
};
// COMPACT form of a static line (32 B, same index as FastKeyInst[]): all a static batch needs when none of its programs
// installs more than two triggers -- the shape of every two-term rule -- so that an install batch reads 2 KB instead
// of 4 KB and two 16-byte loads per lane instead of four.  Result handle, format and variables of a program that
// completes at once (FSM_RESULT_NOW) are read from its FastKeyInst line.
struct FastStatic			// 32 B
{
	uint32_t ev0, info0;		// first installed trigger; info: signal byte as in a bucket entry (8) | variable (8) << 8 | bucket (4) << 16 | rank in the
	uint32_t ev1, info1;		// batch's entries of that bucket (8) << 20 | FSI_LAST | FSI_PRESENT | template slot (2) << 30
	uint32_t hw0;			// as FastKeyInst::hw0
	uint32_t misc;			// range (6) | expRank (8) << 6 | expClose (8) << 14 | FSM_* flags
	uint32_t ranks;			// resRank (8) | dispRank (8) << 8 | items of the batch's immediate results (8) << 16
	uint32_t totals;		// as FastKeyInst::totals
};
enum {
	FSI_BUCKET_SHIFT=16, FSI_RANK_SHIFT=20, FSI_LAST=1u<<28, FSI_PRESENT=1u<<29, FSI_SLOT_SHIFT=30,
	FSM_RANGE_MASK=0x3Fu, FSM_EXPRANK_SHIFT=6, FSM_EXPCLOSE_SHIFT=14,
	FSM_END_SET=1u<<22, FSM_START_SET=1u<<23, FSM_RESULT_NOW=1u<<24, FSM_DISPOSE_NOW=1u<<25, FSM_BATCH_COMPACT=1u<<26
};

// rule word of the fast tier (LDS, u32 per rule instance)
enum {
	H_VALUE_MASK=0xFu, H_COUNT_SHIFT=4, H_COUNT_MASK=0x1Fu, H_END_SHIFT=9, H_END_MASK=0xFFu, H_ENDZERO=1u<<17,
	H_DONE=1u<<18, H_ACTIVE=1u<<19, H_TMASK_SHIFT=20, H_TMASK_MASK=0x7u, H_NITEMS_SHIFT=23, H_NITEMS_MASK=0x3u, H_HASSTART=1u<<25,
	H_COLD=1u<<26,			// start of the match and captured items are in the rule's record in HBM (else: the key lexem + the static line)
	H_LISTED=1u<<27,		// already in the dispose list of the current transition
	H_VISIBLE=1u<<28,		// the pattern has a result handle: completing the rule emits a result (static lines only; else the record tells)
	H_MARK=1u<<31			// set while a lane of fireBatch works on the rule: a second hit on the same rule in one scan step sees it
};

// ---- a fresh slot while its program is being installed: the alternative-key replay and the key triggers fire before
// anything is stored.  Shared by the kernel (dynamic batches) and the host (static lines).
enum {S_HASSTART=1u, S_DONE=2u, S_FIN=4u, S_DEL=8u, S_ODD=16u, S_RESULT=32u, S_HASLIST=64u, S_TOOK=128u};
struct Sim
{
	uint32_t value, count, end, startLex, nItems, it0, it1, it2, nFires, flags;
	// S_HASLIST: the result shares the rule's item list only if the list existed when the rule matched (cpp:941-953):
	// items captured later join that list (and show in the result), or start a list the result never sees
};
SPA_HD static inline void fireLocal( Sim& s, uint32_t info, uint32_t esord, uint32_t elex, uint32_t withItems)
{
	const uint32_t sigtype = (info >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK, sigval = info & FTI_SIGVAL_MASK;
	uint32_t m = 0, f = 0, took = 0;
	s.nFires += 1;
	if (sigtype == SIG_ANY)
	{
		took = 1;
		if (s.count > 0) { m = 1; s.count -= 1; f = (s.count == 0); if (s.end < esord+1) s.end = esord+1; }
	}
	else if (sigtype == SIG_SEQUENCE || sigtype == SIG_SEQUENCE_IMM)
	{
		if (sigval == s.value && (sigtype == SIG_SEQUENCE ? (s.end <= esord) : (s.end == esord)))
		{
			s.end = esord+1; s.value = sigval-1;
			if (s.count > 0) { s.count -= 1; m = (s.count == 0); } else m = 1;
			f = (s.value == 0); took = 1;
		}
	}
	else if (sigtype == SIG_WITHIN)
	{
		if ((sigval & s.value) != 0 && s.end <= esord)
		{
			s.end = esord+1; s.value &= ~sigval;
			if (s.count > 0) { s.count -= 1; m = (s.count == 0); } else m = 1;
			took = 1;
		}
	}
	else	// SIG_DEL: the rule goes to the dispose list, nothing else happens (cpp:868-876)
	{
		s.count = 0; s.value = 0; s.flags |= S_DEL;
		return;
	}
	if (took)
	{
		if ((info & FTI_HASVAR) && withItems)
		{
			const uint32_t item = elex | ((info >> FTI_VAR_SHIFT) << 24);
			if ((s.flags & S_DONE) && !(s.flags & S_HASLIST)) {}
			else if (s.nItems == 0) { s.it0 = item; s.nItems = 1; }
			else if (s.nItems == 1) { s.it1 = item; s.nItems = 2; }
			else if (s.nItems == 2) { s.it2 = item; s.nItems = 3; }
			else s.flags |= S_ODD;
		}
		if (!(s.flags & S_HASSTART)) { s.startLex = elex; s.flags |= S_TOOK; if (esord) s.flags |= S_HASSTART; }
	}
	if (m)
	{
		if (!(s.flags & S_DONE)) { s.flags |= S_DONE | S_RESULT; if (s.nItems) s.flags |= S_HASLIST; }
		if (f) s.flags |= S_FIN;
	}
}

// hash table over key events and stop words (open addressing, keyHash of l2_tables.h)
struct FastKeyEntry			// 16 B
{
	uint32_t event;
	uint32_t kiBegin;		// FastKeyInst[kiBegin .. kiBegin+kiCount)
	uint32_t kiCount;
	uint32_t stopIdx;		// 1-based slot in the per-document stop-word log, 0 = not a stop word
};

// ---- per-wave LDS image: static layout inside the kernel (struct LdsDoc of l2_fast_kernel.hip), one kernel
// instance per capacity pair.  R = rule instances whose hot state is in LDS, T = trigger-bucket entries in LDS.
// Rule ids >= R and bucket positions beyond a bucket's LDS region live in the wave's spill area in HBM, so a
// burst (a frequent word that keys hundreds of programs) slows a document down instead of failing it.
enum {FAST_LISTCAP=64, FAST_EXPCAP=1024, FAST_MAXSTOP=64, FAST_SPILL_BUCKET=1024, FAST_VARIANTS=5, FAST_RQCAP=192};

struct FastSpillLayout			// per-wave spill + cold area in HBM, offsets in u32 words
{
	uint32_t maxRules;		// total rule ids (LDS + spill), <= 4096
	uint32_t oCold;			// u32[8*maxRules]  {resultHandle, formatHandle, first taken lexem, item0, item1, item2, -, -}, item = lexem | variable<<24:
					// written when the rule is installed / takes an event, read when it matches after its installation
	uint32_t oHot, oLink;	// spill rules (ids R..maxRules): same shapes as in LDS (u32 per element here)
	uint32_t oKi, oLex0;		// ... their install line and key lexem
	uint32_t oFree;			// u32[maxRules]  stack of free spill ids
	uint32_t oEnt;			// {event, ts} per spill bucket entry: 16 rows of FAST_SPILL_BUCKET
	uint32_t oStaged;		// staged results, 8 words each
	uint32_t maxStaged;
	uint32_t oList;			// u32[maxRules] long dispose lists
	uint32_t oExp;			// u32[W][maxRules] expiry rows beyond their LDS part (W = 1 << expShift)
	uint32_t totalWords;
};

// per-document status of the fast tier beyond SPD_* of l2_device.h: the document is handed to the general kernel
enum {SPD_FAST_FALLBACK=100};

struct FastParams
{
	const FastKeyInst* keyinst;
	const FastStatic* statics;	// compact static lines, same index
	const FastKeyEntry* keytab;
	uint32_t keymask;
	uint32_t nofStopWords;
	// input
	const uint32_t* lexems;		// sp_lexem_t[]: id, ordpos, origpos, origsize
	const uint32_t* origseg;	// optional
	const uint64_t* docOffsets;	// ndocs+1 lexem indices, or NULL when docRangesIn is given
	const uint64_t* docRangesIn;	// ndocs x (first lexem, count)
	uint32_t ndocs;
	uint32_t withItems;
	// working memory
	uint32_t expShift;		// W = 1 << expShift expiry rows: the smallest power of two above the largest position range
	uint32_t bucketMeta[ 16];	// LDS region of each trigger bucket: first entry | capacity << 16 (sized from the rule set)
	FastSpillLayout spill;
	uint32_t* spillBase;		// per wave: spill.totalWords
	uint32_t* docCursor;
	// output (same buffers and formats as the general kernel)
	uint64_t* counters;		// SPC_*
	uint32_t* results; uint64_t resultCapacity;
	uint32_t* items; uint64_t itemCapacity;
	uint64_t* docRange; uint64_t* docStats; int32_t* docStatus;
	uint32_t withFormats; uint32_t* resultFormat; uint32_t* itemFormat;
	// documents the fast tier hands to the general kernel: fallbackList[ atomicAdd( fallbackCount)]
	uint32_t* fallbackList; uint32_t* fallbackCount;
	uint32_t* diag;			// [16] hand-overs in all / by reason (FB_* of l2_fast_kernel.hip), may be NULL
	uint64_t* prof;			// [8] wave-cycles per phase (make PROF=1 builds), may be NULL
};

} // namespace
#endif
