// Level-2 rule automaton, fast tier (gfx950): one wavefront per document, the document's whole hot
// state in LDS.
//
// What it replaces (reference, CPU): the same functions as l2_kernel.hip -- StateMachine::doTransition /
// fireSignal / installProgram / setCurrentPos / replayPastEvent and PatternMatcherContext::putInput /
// fetchResults (src/ruleMatcherAutomaton.cpp:672-1334, src/patternMatcher.cpp:131-301) -- for FLAT rule
// sets (l2_fast.h: every program has at most 3 triggers over input terms, range <= 63, no rule listens
// to another rule's result).  Results, their order, the captured items and the statistics are those of
// the reference; the general kernel (l2_kernel.hip) stays the engine for everything else and for the
// documents this one hands over (SPD_FAST_FALLBACK).
//
// State of one document (per wave):
//   LDS   rule word u32 per rule instance {value:4 count:5 end_ordpos:8+1 done active trigMask:3 nItems:2 hasStart},
//         the bucket position of each of its <= 3 installed triggers (u16), its link in the expiry list of its
//         position (u16); the 16 trigger buckets of the reference's EventTriggerTable (cpp:114-257) as
//         {event, trigger id | signal byte | variable} entries, each bucket a fixed region sized by the host from
//         the rule set, in exactly the reference's positions (append at the end, swap-with-last removal); the 64
//         expiry list heads; the stop-word log {lexem index, ordpos, timestamp}; the dispose list.
//         The layout is static (one kernel instance per capacity pair): every access is a ds instruction with an
//         immediate offset.
//         Since round 3 also the rule's install line (index into FastKeyInst[]) and its key lexem: what a rule has
//         captured when it is installed -- result handle, format, start of the match, the key trigger's item -- follows
//         from those two, so nothing is written to HBM for a rule that expires untouched (85 % of them).
//   HBM   a 32-byte record {result handle, format, first taken lexem, <= 3 captured items as lexem index | variable}
//         ONLY for a rule that takes an event without completing (H_COLD) or was installed by a dynamic batch; staged
//         results (32 B) expanded to sp_result_t / sp_result_item_t records at the document's end; a spill area for
//         rule ids and bucket positions beyond the LDS capacities.
// Two instances of every step: the normal one touches LDS only; while a burst (a frequent word that keys
// hundreds of programs) has rule ids or bucket entries in the spill area the document runs the instance whose
// accessors look at both places (wave-uniform switch, checked when the position advances).
// A rule's end_ordpos is kept modulo 256: it lies within 63 positions of the current one.
//
// Integer only, no MFMA.  Control flow is wave-uniform; lanes are workers where the algorithm has width:
// scanning a bucket for an event (the 64-lane form of the SSE scan, cpp:179-226), installing the <= 64
// programs of a key event at once (the key trigger and an alternative-key replay are fired in registers
// before anything is stored), removing the triggers of up to 64 rules at once (removals in different
// buckets commute; inside a bucket they run in list order), fetching lexems, writing results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2_fast.h"
#include "l2_device.h"
#include "wave_scan.h"

using namespace spa;

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;
typedef uint8_t u8;

#define LANE ((u32)(threadIdx.x & 63u))
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(4))) FastParams& KP;
__device__ __forceinline__ KP kernelParams() { return *(const __attribute__((address_space(4))) FastParams*)__builtin_amdgcn_kernarg_segment_ptr(); }

// cross-lane hand-over through LDS / the wave's own spill area: LDS and vector memory operations of one
// wave execute in order; the fence keeps the compiler from moving or reusing accesses across the point
#define WAVE_FENCE() __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront")

enum {NIL16=0xFFFFu};

// Phase profile (make PROF=1): wave-cycles per phase summed into P.prof[0..7]
// 0 bucket scan + fires, 1 installs, 2 deactivation of finished rules, 3 expiry, 4 results at the document end, 5 lexem fetch + key probes
#ifdef SPA_PROF
#define PROF_DECL u64 prof_t0 = __builtin_amdgcn_s_memtime()
#define PROF_ADD( SLOT) do { const u64 prof_t1 = __builtin_amdgcn_s_memtime(); w.prof[ SLOT] += prof_t1 - prof_t0; prof_t0 = prof_t1; } while (0)
#else
#define PROF_DECL do {} while (0)
#define PROF_ADD( SLOT) do {} while (0)
#endif

__device__ __forceinline__ u32 bcast0( u32 v) { return __builtin_amdgcn_readfirstlane( v); }
__device__ __forceinline__ u64 lanesBelow() { return (1ull << LANE) - 1ull; }
__device__ __forceinline__ u32 evhash( u32 a)		// src/ruleMatcherAutomaton.cpp:34-40
{
	a += ~(a>>5);
	a +=  (a<<3);
	a ^=  (a>>4);
	return a;
}
__device__ __forceinline__ uint4 ld4( const void* p) { const u32x4 v = *(const u32x4*)p; return make_uint4( v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4( void* p, u32 a, u32 b, u32 c, u32 d) { u32x4 v; v.x = a; v.y = b; v.z = c; v.w = d; *(u32x4*)p = v; }
__device__ __forceinline__ u32 ldu( const u32* p) { return __builtin_amdgcn_readfirstlane( *p); }
__device__ __forceinline__ uint4 ldu4( const void* p)
{
	uint4 v = ld4( p);
	v.x = bcast0( v.x); v.y = bcast0( v.y); v.z = bcast0( v.z); v.w = bcast0( v.w);
	return v;
}
__device__ __forceinline__ u32 waveScanOr( u32 v)		// inclusive, all 64 lanes active
{
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x111, 0xF, 0xF, true);
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x112, 0xF, 0xF, true);
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x114, 0xF, 0xF, true);
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x118, 0xF, 0xF, true);
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x142, 0xA, 0xF, false);
	v |= (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x143, 0xC, 0xF, false);
	return v;
}
__device__ __forceinline__ u32 fromLaneBelow( u32 v)		// value of lane-1 (0 in lane 0)
{
	const u32 t = (u32)__builtin_amdgcn_ds_bpermute( (int)((LANE - 1u) << 2), (int)v);
	return LANE ? t : 0u;
}
__device__ __forceinline__ u32 byteField( u32 c0, u32 c1, u32 c2, u32 c3, u32 h)
{
	const u32 wsel = h >> 2;
	const u32 v = wsel == 0 ? c0 : wsel == 1 ? c1 : wsel == 2 ? c2 : c3;
	return (v >> ((h & 3u)*8)) & 0xFFu;
}
__device__ __forceinline__ void byteInc( u32& c0, u32& c1, u32& c2, u32& c3, u32 h)
{
	// (selects, not an if-chain: the compiler turns the chain into an indexed array in scratch memory)
	const u32 inc = 1u << ((h & 3u)*8), ws = h >> 2;
	c0 += ws == 0 ? inc : 0u; c1 += ws == 1 ? inc : 0u; c2 += ws == 2 ? inc : 0u; c3 += ws == 3 ? inc : 0u;
}

// the document leaves the fast tier (a capacity of the spill area, or a case the compact state cannot express)
#define FALLBACK( WHY) do { w.err = SPD_FAST_FALLBACK; w.why = (WHY); } while (0)
enum {FB_LEXEMS=1, FB_DISPOSE=2, FB_ITEM_AFTER_RESULT=3, FB_ITEMS=4, FB_STAGED=5, FB_RULES=8, FB_BUCKET_SIZE=9, FB_POSLEXEMS=10};

// end_ordpos <= pos (sequence / within guard) and == pos (sequence_imm), from the 8 low bits kept in the word;
// the true value lies in [pos-63, pos+1]
__device__ __forceinline__ bool endLE( u32 hw, u32 pos) { return (hw & H_ENDZERO) || ((hw >> H_END_SHIFT) & H_END_MASK) != ((pos + 1u) & 0xFFu); }
__device__ __forceinline__ bool endEQ( u32 hw, u32 pos) { return (hw & H_ENDZERO) ? (pos == 0) : (((hw >> H_END_SHIFT) & H_END_MASK) == (pos & 0xFFu)); }
__device__ __forceinline__ u32 withEnd( u32 hw, u32 end) { return (hw & ~((H_END_MASK << H_END_SHIFT) | H_ENDZERO)) | ((end & 0xFFu) << H_END_SHIFT); }

// ---------------------------------------------------------------- LDS image of a document (static layout)
// (LDS is what bounds the number of documents in flight per CU, and the kernel is latency bound: every byte counts.
//  The stop-word log lives in three vector registers, lane = stop word.)
template <bool SMALL> struct FreeId { typedef u8 type; };
template <> struct FreeId<false> { typedef u16 type; };
template <int R, int T>
struct LdsDoc
{
	u32 hot[ R];			// rule word
	u32 kl[ R];			// install line of the rule (FastKeyInst index, 20 bits) | its key lexem (index inside the document, low 12 bits) << 20
	u64 ent[ T];			// bucket entries: event id | (trigger id (rule<<2 | slot) | signal byte << 16 | variable << 24) << 32; bucket h owns [base_h, base_h + cap_h)
	u32 bsize[ 16];			// bucket sizes
	u32 bmeta[ 16];			// base_h | cap_h << 16
	u16 link[ 3*R];			// bucket << 12 | position of trigger slot j of rule r at [3r+j]
	typename FreeId<(R <= 256)>::type freeS[ R];	// stack of free rule ids < R
	u16 exp[ FAST_EXPCAP];		// expiry lists: row (position & (W-1)) holds the rules that expire at that position, in definition order
	u16 expCnt[ 64];		// their lengths
	u16 list[ FAST_LISTCAP];	// dispose list of the current transition
	u16 rq[ FAST_RQCAP];		// removal queue: the triggers (rule << 2 | slot) of a deactivation batch grouped by bucket, in removal order
};

// per-document scalars (wave-uniform registers)
struct Wave
{
	u32* sp;		// spill + cold area in HBM
	u32 curpos, timestamp, nInstalled, nAlt, nSignals, nTrig, openLo, openHi;
	u32 freeN, usedL, sFreeN, usedS;	// rule ids: LDS free stack / bump, spill free stack / bump
	u32 nDispose, nStaged, nStagedItems, err;
	u32 lbase;				// lexem index of the event being processed, relative to the document
	u32 spill;				// something of this document lives in the spill area: the two-place accessors run
	u32 why;				// reason of a hand-over (diagnostics)
	u32 posLexems;				// lexems at the current position (the key lexem of a rule is kept modulo 4096)
	// per-lane values (not uniform)
	u32 stopLex, stopOrd, stopTs;		// stop-word log {lexem index, ordpos, timestamp}: lane = stop word index - 1
	u32 bsizeV, bmetaV;			// lane b < 16: size and region of bucket b (mirrors of L.bsize / L.bmeta for uniform reads without an LDS round trip)
	u32 expCntV;				// lane = expiry row: its length (mirror of L.expCnt)
#ifdef SPA_PROF
	u64 prof[ 12];
#endif
};

template <int R, int T>
struct Engine
{
typedef LdsDoc<R,T> Lds;
typedef __attribute__((address_space(3))) Lds& LR;

// ---- rule fields: ids < R in LDS, others in the spill area (same shapes).  SP=false: LDS only.
template <bool SP> static __device__ __forceinline__ u32 ldHot( LR L, const Wave& w, KP P, u32 r) { if (!SP || r < (u32)R) return L.hot[ r]; return w.sp[ P.spill.oHot + (r - R)]; }
template <bool SP> static __device__ __forceinline__ void stHot( LR L, const Wave& w, KP P, u32 r, u32 v) { if (!SP || r < (u32)R) L.hot[ r] = v; else w.sp[ P.spill.oHot + (r - R)] = v; }
template <bool SP> static __device__ __forceinline__ u32 ldLink( LR L, const Wave& w, KP P, u32 r, u32 j) { if (!SP || r < (u32)R) return (u32)L.link[ 3*r + j]; return w.sp[ P.spill.oLink + 3*(r - R) + j]; }
template <bool SP> static __device__ __forceinline__ void stLink( LR L, const Wave& w, KP P, u32 r, u32 j, u32 v) { if (!SP || r < (u32)R) L.link[ 3*r + j] = (u16)v; else w.sp[ P.spill.oLink + 3*(r - R) + j] = v; }
template <bool SP> static __device__ __forceinline__ void ldLinks( LR L, const Wave& w, KP P, u32 r, u32& l0, u32& l1, u32& l2)
{
	if (!SP || r < (u32)R) { l0 = (u32)L.link[ 3*r]; l1 = (u32)L.link[ 3*r+1]; l2 = (u32)L.link[ 3*r+2]; }
	else { const u32* q = &w.sp[ P.spill.oLink + 3*(r - R)]; l0 = q[0]; l1 = q[1]; l2 = q[2]; }
}
// install line and key lexem of a rule in one word: line index (< 2^20, checked by the host) | lexem index modulo 4096 << 20.
// A rule lives at most 63 positions and a position with more than 60 lexems hands the document over (FB_POSLEXEMS),
// so the key lexem lies less than 4096 lexems behind the current one.
template <bool SP> static __device__ __forceinline__ u32 ldKl( LR L, const Wave& w, KP P, u32 r) { if (!SP || r < (u32)R) return L.kl[ r]; return w.sp[ P.spill.oKi + (r - R)]; }
template <bool SP> static __device__ __forceinline__ void stKl( LR L, const Wave& w, KP P, u32 r, u32 ki, u32 lx)
{ const u32 v = ki | (lx << 20); if (!SP || r < (u32)R) L.kl[ r] = v; else w.sp[ P.spill.oKi + (r - R)] = v; }
static __device__ __forceinline__ u32 keyLexemOf( u32 kl, u32 lbase) { return lbase - ((lbase - (kl >> 20)) & 0xFFFu); }
// ---- expiry rows: entry i of row `row` sits in LDS while i < C = FAST_EXPCAP >> expShift, else in the row's spill part
template <bool SP> static __device__ __forceinline__ u32 ldExp( LR L, const Wave& w, KP P, u32 row, u32 i)
{ const u32 C = (u32)FAST_EXPCAP >> P.expShift; if (!SP || i < C) return (u32)L.exp[ row*C + i]; return w.sp[ P.spill.oExp + row*P.spill.maxRules + (i - C)]; }
template <bool SP> static __device__ __forceinline__ void stExp( LR L, const Wave& w, KP P, u32 row, u32 i, u32 r)
{ const u32 C = (u32)FAST_EXPCAP >> P.expShift; if (!SP || i < C) L.exp[ row*C + i] = (u16)r; else w.sp[ P.spill.oExp + row*P.spill.maxRules + (i - C)] = r; }
// ---- bucket entries: position p of bucket h sits at base_h + p while p < cap_h, else in the bucket's spill row
template <bool SP> static __device__ __forceinline__ u32 ldEv( LR L, const Wave& w, KP P, u32 h, u32 meta, u32 pos)
{ if (!SP || pos < (meta >> 16)) return (u32)L.ent[ (meta & 0xFFFFu) + pos]; return w.sp[ P.spill.oEnt + 2*(h*FAST_SPILL_BUCKET + pos - (meta >> 16))]; }
template <bool SP> static __device__ __forceinline__ uint2 ldEnt( LR L, const Wave& w, KP P, u32 h, u32 meta, u32 pos)
{ if (!SP || pos < (meta >> 16)) { const u64 v = L.ent[ (meta & 0xFFFFu) + pos]; return make_uint2( (u32)v, (u32)(v >> 32)); } return *(const uint2*)&w.sp[ P.spill.oEnt + 2*(h*FAST_SPILL_BUCKET + pos - (meta >> 16))]; }
template <bool SP> static __device__ __forceinline__ u32 ldTs( LR L, const Wave& w, KP P, u32 h, u32 meta, u32 pos) { return ldEnt<SP>( L, w, P, h, meta, pos).y; }
template <bool SP> static __device__ __forceinline__ void stEnt( LR L, const Wave& w, KP P, u32 h, u32 meta, u32 pos, u32 e, u32 t)
{
	if (!SP || pos < (meta >> 16)) L.ent[ (meta & 0xFFFFu) + pos] = (u64)e | ((u64)t << 32);
	else *(uint2*)&w.sp[ P.spill.oEnt + 2*(h*FAST_SPILL_BUCKET + pos - (meta >> 16))] = make_uint2( e, t);
}
// ---- the dispose list: first FAST_LISTCAP entries in LDS, the rest in the spill area
template <bool SP> static __device__ __forceinline__ u32 ldList( LR L, const Wave& w, KP P, u32 i) { if (!SP || i < (u32)FAST_LISTCAP) return (u32)L.list[ i]; return w.sp[ P.spill.oList + i]; }
static __device__ __forceinline__ void pushList( LR L, Wave& w, KP P, u32 i, u32 r)	// uniform index, one lane stores
{
	if (i < (u32)FAST_LISTCAP) { if (LANE == 0) L.list[ i] = (u16)r; }
	else { w.spill = 1; if (LANE == 0) w.sp[ P.spill.oList + i] = r; }
}

// ---------------------------------------------------------------- staged results
// {resultHandle, formatHandle, first lexem, last lexem, nItems | var0<<8 | var1<<16 | var2<<24, item lexems x3} (items latest first)
// or, for a rule that completes with what its static install line tells (no read of that line on the way: the epilogue reads it,
// 64 results at a time): {STAGED_LINE | install line, -, first lexem, last lexem, static items (2) | new item (1) << 2 | its variable << 8, key lexem}
enum {STAGED_LINE=0x80000000u};
static __device__ __forceinline__ void stageLineResult( Wave& w, KP P, u32 at, u32 ki, u32 startLex, u32 endLex, u32 nStatic, u32 hasNew, u32 newVar, u32 keyLex)
{
	u32* S = &w.sp[ P.spill.oStaged + 8*at];
	st4( S, (u32)STAGED_LINE | ki, 0u, startLex, endLex);
	*(uint2*)(S+4) = make_uint2( nStatic | (hasNew << 2) | (newVar << 8), keyLex);
}
static __device__ __forceinline__ void stageResult( Wave& w, KP P, u32 at, u32 handle, u32 fmt, u32 startLex, u32 endLex, u32 nItems, u32 vars, u32 i0, u32 i1, u32 i2)
{
	u32* S = &w.sp[ P.spill.oStaged + 8*at];
	st4( S, handle, fmt, startLex, endLex);
	st4( S+4, nItems | (vars << 8), i0, i1, i2);
}

// ---------------------------------------------------------------- fireSignal (cpp:772-979) on an installed trigger
// uniform: every lane computes the same; stores by lane 0.
// What the rule has captured so far (start of the match, items) is not stored anywhere while it is what the install
// line says for the key lexem (!H_COLD): it is rebuilt here when the rule completes -- one read of the line, which the
// result needs anyway for its handle -- and written to the rule's record in HBM only if the rule takes this event and
// goes on waiting (programs of three triggers, cardinalities).
template <bool SP>
static __device__ __forceinline__ void fireSignal( LR L, Wave& w, KP P, u32 tsv, u32 sord)
{
	const u32 tid = tsv & 0xFFFFu, r = tid >> 2;
	const u32 sigval = (tsv >> 16) & 0xFu, sigtype = (tsv >> 20) & 0x7u, hasVar = (tsv >> 23) & 1u, variable = tsv >> 24;
	u32 hw = bcast0( ldHot<SP>( L, w, P, r));
	w.nSignals += 1;
	u32 value = hw & H_VALUE_MASK, count = (hw >> H_COUNT_SHIFT) & H_COUNT_MASK;
	bool match = false, take = false, fin = false;
	switch (sigtype)
	{
		case SIG_ANY:
			take = true;
			if (count > 0) { match = true; --count; fin = (count == 0); hw = withEnd( hw, sord+1); }	// end_ordpos <= curpos+1 always: the maximum is the event's end
			break;
		case SIG_SEQUENCE:
		case SIG_SEQUENCE_IMM:
			if (sigval == value && (sigtype == SIG_SEQUENCE ? endLE( hw, sord) : endEQ( hw, sord)))
			{
				hw = withEnd( hw, sord+1); value = sigval-1;
				if (count > 0) { --count; match = (count == 0); } else match = true;
				fin = (value == 0); take = true;
			}
			break;
		case SIG_WITHIN:
			if ((sigval & value) != 0 && endLE( hw, sord))
			{
				hw = withEnd( hw, sord+1); value &= ~sigval;
				if (count > 0) { --count; match = (count == 0); } else match = true;
				take = true;		// (never finished: the mask of a within slot keeps its upper bits, cpp:585)
			}
			break;
		default: // SIG_DEL
			hw &= ~(H_VALUE_MASK | (H_COUNT_MASK << H_COUNT_SHIFT));
			if (!(hw & H_LISTED))
			{
				// (a rule is listed once per transition: of two entries only the first would act, cpp:679-702)
				if (w.nDispose < P.spill.maxRules) { pushList( L, w, P, w.nDispose, r); w.nDispose += 1; hw |= H_LISTED; } else FALLBACK( FB_DISPOSE);
			}
			if (LANE == 0) stHot<SP>( L, w, P, r, hw);
			return;
	}
	const bool done = (hw & H_DONE) != 0;
	u32 nItems = (hw >> H_NITEMS_SHIFT) & H_NITEMS_MASK;
	bool newItem = false, newStart = false;
	if (take)
	{
		if (hasVar && P.withItems)
		{
			if (done)
			{
				// an item captured after the rule's result exists: it would join the result's list if that list
				// existed when the result was made (cpp:941-953 share the reference) -- not expressible here
				if (nItems) { FALLBACK( FB_ITEM_AFTER_RESULT); return; }
			}
			else if (nItems < 3u) newItem = true;
			else { FALLBACK( FB_ITEMS); return; }
		}
		newStart = !(hw & H_HASSTART);
	}
	if (!done && match && !(hw & H_COLD))
	{
		// ---- a rule completes with what its install line tells for its key lexem (+ the event that completes it): nothing is read here
		if (hw & H_VISIBLE)
		{
			if (w.nStaged < P.spill.maxStaged)
			{
				const u32 kl = bcast0( ldKl<SP>( L, w, P, r)), lx = keyLexemOf( kl, w.lbase);
				if (LANE == 0) stageLineResult( w, P, w.nStaged, kl & 0xFFFFFu, newStart ? w.lbase : lx, w.lbase, nItems, newItem ? 1u : 0u, variable, lx);
				w.nStaged += 1; w.nStagedItems += nItems + (newItem ? 1u : 0u);
			}
			else FALLBACK( FB_STAGED);
		}
		if (newItem) ++nItems;
	}
	else if (!done && (match || newItem || newStart))
	{
		// ---- what the rule has captured: its record, or the key lexem + the install line
		u32* cold = &w.sp[ P.spill.oCold + 8*r];		// {resultHandle, formatHandle, first taken lexem, item0, item1, item2, -, -}; item = lexem | variable<<24
		u32 handle, fmt, startLex, it0, it1, it2;
		if (hw & H_COLD)
		{
			const uint4 c0 = ldu4( cold), c1 = ldu4( cold + 4);		// written by this wave (same-wave store -> load)
			handle = c0.x; fmt = c0.y; startLex = c0.z; it0 = c0.w; it1 = c1.x; it2 = c1.y;
		}
		else
		{
			const u32 kl = bcast0( ldKl<SP>( L, w, P, r)), lx = keyLexemOf( kl, w.lbase);
			const u32* K = (const u32*)&P.keyinst[ kl & 0xFFFFFu];
			handle = ldu( K); fmt = ldu( K+1);
			const u32 vars = ldu( K+14);
			startLex = lx;		// (a static line is installed at a position != 0: a key fire that took set the start)
			it0 = lx | ((vars & 0xFFu) << 24); it1 = lx | (((vars >> 8) & 0xFFu) << 24); it2 = lx | (((vars >> 16) & 0xFFu) << 24);
		}
		if (newItem)
		{
			const u32 item = w.lbase | (variable << 24);
			if (nItems == 0) it0 = item; else if (nItems == 1) it1 = item; else it2 = item;
			++nItems;
		}
		if (newStart) startLex = w.lbase;
		if (match)
		{
			if (handle)
			{
				if (w.nStaged < P.spill.maxStaged)
				{
					// items latest first
					const u32 ia = nItems == 3 ? it2 : nItems == 2 ? it1 : it0;
					const u32 ib = nItems == 3 ? it1 : it0;
					const u32 ic = it0;
					const u32 vars = nItems == 0 ? 0u : nItems == 1 ? (ia >> 24) : nItems == 2 ? ((ia >> 24) | ((ib >> 24) << 8)) : ((ia >> 24) | ((ib >> 24) << 8) | ((ic >> 24) << 16));
					if (LANE == 0) stageResult( w, P, w.nStaged, handle, fmt, startLex, w.lbase, nItems, vars, ia & 0xFFFFFFu, ib & 0xFFFFFFu, ic & 0xFFFFFFu);
					w.nStaged += 1; w.nStagedItems += nItems;
				}
				else FALLBACK( FB_STAGED);
			}
		}
		else
		{
			// the rule goes on waiting with more than its install line tells: its record takes over
			if (LANE == 0) { st4( cold, handle, fmt, startLex, it0); *(uint2*)(cold + 4) = make_uint2( it1, it2); }
			hw |= H_COLD;
			WAVE_FENCE();
		}
	}
	if (newStart && sord) hw |= H_HASSTART;		// (a start at ordinal position 0 counts as unset, cpp:919)
	hw = (hw & ~(H_VALUE_MASK | (H_COUNT_MASK << H_COUNT_SHIFT) | (H_NITEMS_MASK << H_NITEMS_SHIFT))) | value | (count << H_COUNT_SHIFT) | (nItems << H_NITEMS_SHIFT);
	if (match && !done) hw |= H_DONE;
	if (match && fin && !(hw & H_LISTED))
	{
		if (w.nDispose < P.spill.maxRules) { pushList( L, w, P, w.nDispose, r); w.nDispose += 1; hw |= H_LISTED; } else FALLBACK( FB_DISPOSE);
	}
	if (LANE == 0) stHot<SP>( L, w, P, r, hw);
}

// ---------------------------------------------------------------- fireSignal for all hits of a bucket scan step at once
// The hits of one step are triggers of different rules almost always: then the fires do not see each other, and what they share
// -- the staged results, the dispose list, the counters -- is placed by ballot rank in hit order.  Lane = hit; the same statements
// as fireSignal per lane.  Returns false with nothing changed when two hits of the step belong to one rule (programs with two
// equal terms) or the dispose list would leave LDS: the caller fires them one after the other.  LDS-only instance.
static __device__ __forceinline__ bool fireBatch( LR L, Wave& w, KP P, const u64 m, const u32 tsv, const u32 sord)
{
	const bool hit = ((m >> LANE) & 1ull) != 0;
	const u32 nh = (u32)__popcll( m);
	if (w.nDispose + nh > (u32)FAST_LISTCAP) return false;
	const u32 tid = tsv & 0xFFFFu, r = hit ? (tid >> 2) : 0u;
	const u32 sigval = (tsv >> 16) & 0xFu, sigtype = (tsv >> 20) & 0x7u, hasVar = (tsv >> 23) & 1u, variable = tsv >> 24;
	u32 hw = 0;
	if (hit) hw = __hip_atomic_fetch_or( &L.hot[ r], (u32)H_MARK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	if (__ballot( hit && (hw & (u32)H_MARK)))
	{
		if (hit) __hip_atomic_fetch_and( &L.hot[ r], ~(u32)H_MARK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		WAVE_FENCE();
		return false;
	}
	u32 value = hw & H_VALUE_MASK, count = (hw >> H_COUNT_SHIFT) & H_COUNT_MASK;
	bool match = false, take = false, fin = false;
	const bool isDel = hit && sigtype != (u32)SIG_ANY && sigtype != (u32)SIG_SEQUENCE && sigtype != (u32)SIG_SEQUENCE_IMM && sigtype != (u32)SIG_WITHIN;
	if (hit && !isDel)
	{
		if (sigtype == (u32)SIG_ANY)
		{
			take = true;
			if (count > 0) { match = true; --count; fin = (count == 0); hw = withEnd( hw, sord+1); }
		}
		else if (sigtype == (u32)SIG_WITHIN)
		{
			if ((sigval & value) != 0 && endLE( hw, sord))
			{
				hw = withEnd( hw, sord+1); value &= ~sigval;
				if (count > 0) { --count; match = (count == 0); } else match = true;
				take = true;
			}
		}
		else if (sigval == value && (sigtype == (u32)SIG_SEQUENCE ? endLE( hw, sord) : endEQ( hw, sord)))
		{
			hw = withEnd( hw, sord+1); value = sigval-1;
			if (count > 0) { --count; match = (count == 0); } else match = true;
			fin = (value == 0); take = true;
		}
	}
	if (isDel) hw &= ~(H_VALUE_MASK | (H_COUNT_MASK << H_COUNT_SHIFT));
	const bool done = (hw & H_DONE) != 0;
	u32 nItems = (hw >> H_NITEMS_SHIFT) & H_NITEMS_MASK;
	bool newItem = false, newStart = false;
	u32 why = 0;
	if (take)
	{
		if (hasVar && P.withItems)
		{
			if (done) { if (nItems) why = FB_ITEM_AFTER_RESULT; }
			else if (nItems < 3u) newItem = true;
			else why = FB_ITEMS;
		}
		newStart = !(hw & H_HASSTART);
	}
	const bool live = hit && !isDel && !done;
	const bool lineCase = live && match && !(hw & H_COLD);					// completes with what its install line tells
	const bool recCase = live && !lineCase && (match || newItem || newStart);		// needs what the rule has captured
	// ---- what the rule has captured: its record, or the key lexem + the install line
	u32 handle = 0, fmt = 0, startLex = 0, it0 = 0, it1 = 0, it2 = 0, kl = 0, lx = 0;
	if ((lineCase && (hw & H_VISIBLE)) || (recCase && !(hw & H_COLD))) { kl = L.kl[ r]; lx = keyLexemOf( kl, w.lbase); }
	u32* cold = &w.sp[ P.spill.oCold + 8*r];
	u32 nItemsOut = nItems;
	if (recCase)
	{
		if (hw & H_COLD)
		{
			const uint4 c0 = ld4( cold); const uint2 c1 = *(const uint2*)(cold + 4);		// written by this wave (same-wave store -> load)
			handle = c0.x; fmt = c0.y; startLex = c0.z; it0 = c0.w; it1 = c1.x; it2 = c1.y;
		}
		else
		{
			const u32* K = (const u32*)&P.keyinst[ kl & 0xFFFFFu];
			handle = K[ 0]; fmt = K[ 1];
			const u32 vars = K[ 14];
			startLex = lx;
			it0 = lx | ((vars & 0xFFu) << 24); it1 = lx | (((vars >> 8) & 0xFFu) << 24); it2 = lx | (((vars >> 16) & 0xFFu) << 24);
		}
		if (newItem)
		{
			const u32 item = w.lbase | (variable << 24);
			if (nItems == 0) it0 = item; else if (nItems == 1) it1 = item; else it2 = item;
			nItemsOut = nItems + 1u;
		}
		if (newStart) startLex = w.lbase;
	}
	else if (lineCase && newItem) nItemsOut = nItems + 1u;
	// ---- results in hit order
	const bool resLine = lineCase && (hw & H_VISIBLE) != 0, resRec = recCase && match && handle != 0;
	const u64 resM = __ballot( resLine || resRec);
	const u32 nres = (u32)__popcll( resM);
	// ---- dispose list in hit order (a rule is listed once per transition)
	const bool listNow = hit && !(hw & H_LISTED) && (isDel || (match && fin));
	const u64 listM = __ballot( listNow);
	const u32 nlist = (u32)__popcll( listM);
	{
		const u64 whyM = __ballot( why != 0);
		u32 fb = 0;
		if (whyM) fb = (u32)__builtin_amdgcn_readlane( why, (u32)__builtin_ctzll( whyM));
		else if (w.nStaged + nres > P.spill.maxStaged) fb = FB_STAGED;
		else if (w.nDispose + nlist > P.spill.maxRules) fb = FB_DISPOSE;
		if (fb) { FALLBACK( fb); return true; }
	}
	w.nSignals += nh;
	if (nres)
	{
		const u32 at = w.nStaged + (u32)__popcll( resM & lanesBelow());
		if (resLine) stageLineResult( w, P, at, kl & 0xFFFFFu, newStart ? w.lbase : lx, w.lbase, nItems, newItem ? 1u : 0u, variable, lx);
		if (resRec)
		{
			// items latest first
			const u32 ia = nItemsOut == 3 ? it2 : nItemsOut == 2 ? it1 : it0;
			const u32 ib = nItemsOut == 3 ? it1 : it0;
			const u32 ic = it0;
			const u32 vars = nItemsOut == 0 ? 0u : nItemsOut == 1 ? (ia >> 24) : nItemsOut == 2 ? ((ia >> 24) | ((ib >> 24) << 8)) : ((ia >> 24) | ((ib >> 24) << 8) | ((ic >> 24) << 16));
			stageResult( w, P, at, handle, fmt, startLex, w.lbase, nItemsOut, vars, ia & 0xFFFFFFu, ib & 0xFFFFFFu, ic & 0xFFFFFFu);
		}
		w.nStaged += nres;
		const u32 itemsIncl = waveScanAdd( (resLine || resRec) ? nItemsOut : 0u);
		w.nStagedItems += (u32)__builtin_amdgcn_readlane( itemsIncl, 63);
	}
	if (recCase && !match)
	{
		// the rule goes on waiting with more than its install line tells: its record takes over
		st4( cold, handle, fmt, startLex, it0); *(uint2*)(cold + 4) = make_uint2( it1, it2);
		hw |= H_COLD;
	}
	if (nlist)
	{
		if (listNow) L.list[ w.nDispose + (u32)__popcll( listM & lanesBelow())] = (u16)r;
		w.nDispose += nlist;
	}
	if (hit)
	{
		if (listNow) hw |= H_LISTED;
		if (!isDel)
		{
			if (newStart && sord) hw |= H_HASSTART;		// (a start at ordinal position 0 counts as unset, cpp:919)
			hw = (hw & ~(H_VALUE_MASK | (H_COUNT_MASK << H_COUNT_SHIFT) | (H_NITEMS_MASK << H_NITEMS_SHIFT))) | value | (count << H_COUNT_SHIFT) | ((live ? nItemsOut : nItems) << H_NITEMS_SHIFT);
			if (match && !done) hw |= H_DONE;
		}
		L.hot[ r] = hw & ~(u32)H_MARK;
	}
	WAVE_FENCE();
	return true;
}

// ---------------------------------------------------------------- deactivation of a list of rules
// deactivateRule (cpp:679-702) for n rules in list order; freeIds: disposeRule (cpp:704-708).  The list is the
// dispose list of the transition (EXPROW = false; a rule is listed once, H_LISTED) or, last defined first, the
// expiry row of a position.
// The only order-dependent part is the swap-with-last removal of the rules' triggers from the 16 buckets
// (cpp:133-152): removals in different buckets do not interact, inside a bucket they must run in list order (rule
// by rule, a rule's triggers last installed first).  So: every lane takes a rule; a prefix count per bucket gives
// every trigger its rank among the batch's removals from its bucket, which is its place in the bucket's part of a
// removal queue in LDS; then LANE b REPLAYS BUCKET b's removals one after the other -- 16 buckets side by side, a
// step is two dependent LDS reads and two stores by one lane.  (Round 2 let every rule's lane wait for its
// trigger's turn: as many rounds of the whole wave as the fullest bucket has removals, 5 per event on the
// pipeline workload -- the sentence delimiter's bucket holds a trigger of every *_struct instance.)
template <bool SP>
static __device__ __forceinline__ void deactivateList( LR L, Wave& w, KP P, const u32 nD, const u32 nE, const u32 row)
{
	// the first nD rules: the dispose list of the transition just done; behind them the nE rules of an expiry row, last defined first.
	// (Nothing looks at the buckets between the end of a transition and the expiry of the next position, so the two lists go through
	// ONE batch: the fixed cost of a batch -- ranks, queue offsets, the replay -- is paid once per event instead of twice.)
	const u32 n = nD + nE;
	for (u32 base=0; base<n && !w.err; base+=64)
	{
		const u32 nb = (n - base) < 64u ? (n - base) : 64u;
		const bool have = LANE < nb;
		const bool isExp = base + LANE >= nD;
		u32 r = 0, hw = 0;
#if defined(SPA_PROF) && !defined(SPA_PROF_INSTALL)
		u64 pd0 = __builtin_amdgcn_s_memtime();
#define PROF_D( SLOT) do { if (__ballot( hw == 0xFFFFFFFFu)) w.err = SPD_ERR_INTERNAL; const u64 pd1 = __builtin_amdgcn_s_memtime(); w.prof[ SLOT] += pd1 - pd0; pd0 = pd1; } while (0)
#else
#define PROF_D( SLOT) do {} while (0)
#endif
		if (have)
		{
			r = isExp ? ldExp<SP>( L, w, P, row, nE - 1u - (base + LANE - nD)) : ldList<SP>( L, w, P, base + LANE);
			hw = ldHot<SP>( L, w, P, r);
		}
		// (a rule of the row that is still active AND listed sits in the dispose list of this very batch: that lane removes its triggers)
		const bool act = have && (hw & H_ACTIVE) && !(isExp && (hw & H_LISTED));
		const u32 mask = act ? ((hw >> H_TMASK_SHIFT) & H_TMASK_MASK) : 0u;
		if (act) stHot<SP>( L, w, P, r, hw & ~(H_ACTIVE | (H_TMASK_MASK << H_TMASK_SHIFT)));
		if (__ballot( mask != 0))
		{
			// my triggers' buckets; their number per bucket as packed byte counters (16 buckets x 8 bits in four words)
			u32 lk[ 3] = {0,0,0};
			if (mask) ldLinks<SP>( L, w, P, r, lk[ 0], lk[ 1], lk[ 2]);
#if defined(SPA_PROF) && !defined(SPA_PROF_INSTALL)
			if (__ballot( lk[ 0] == 0xFFFFFFFFu)) w.err = SPD_ERR_INTERNAL;
#endif
			PROF_D( 6);
			const bool slot2 = __ballot( (mask & 4u) != 0) != 0;		// (two-term rules never use their third trigger slot: its code is skipped wave-wide)
			u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
			for (int j=0; j<2; ++j) if ((mask >> j) & 1u) byteInc( c0, c1, c2, c3, lk[ j] >> 12);
			if (slot2) { if (mask & 4u) byteInc( c0, c1, c2, c3, lk[ 2] >> 12); }
			const u32 i0 = waveScanAdd( c0), i1 = waveScanAdd( c1), i2 = waveScanAdd( c2), i3 = waveScanAdd( c3);	// (fields < 256: 64 x 3)
			const u32 e0 = i0 - c0, e1 = i1 - c1, e2 = i2 - c2, e3 = i3 - c3;
			const u32 t0 = (u32)__builtin_amdgcn_readlane( i0, 63), t1 = (u32)__builtin_amdgcn_readlane( i1, 63);
			const u32 t2 = (u32)__builtin_amdgcn_readlane( i2, 63), t3 = (u32)__builtin_amdgcn_readlane( i3, 63);
			// the queue part of bucket b begins behind the parts of the buckets below it
			const u32 tot = LANE < 16u ? byteField( t0, t1, t2, t3, LANE) : 0u;
			const u32 qoff = waveScanAdd( tot) - tot;
			const u32 total = (u32)__builtin_amdgcn_readlane( qoff + tot, 15);
			// a rule's triggers go last installed first (slot 2, 1, 0)
			if (slot2)
			{
				const u32 h = lk[ 2] >> 12;
				const u32 qo = (u32)__builtin_amdgcn_ds_bpermute( (int)(h << 2), (int)qoff);	// (all lanes: the lanes that hold the offsets must take part)
				if (mask & 4u) L.rq[ qo + byteField( e0, e1, e2, e3, h)] = (u16)(4u*r + 2u);
			}
#pragma unroll
			for (int j=1; j>=0; --j)
			{
				const u32 h = lk[ j] >> 12;
				const u32 qo = (u32)__builtin_amdgcn_ds_bpermute( (int)(h << 2), (int)qoff);
				if ((mask >> j) & 1u)
				{
					u32 mineBefore = 0;
					if (slot2 && (mask & 4u) && (lk[ 2] >> 12) == h) ++mineBefore;
					if (j == 0 && (mask & 2u) && (lk[ 1] >> 12) == h) ++mineBefore;
					L.rq[ qo + byteField( e0, e1, e2, e3, h) + mineBefore] = (u16)(4u*r + (u32)j);
				}
			}
			WAVE_FENCE();
			PROF_D( 7);
			if (LANE < 16u && tot)
			{
				// (a register replay of four removals per step -- trigger ids, positions and the bucket's last four entries in
				// two rounds of loads -- was measured: slower, 184.6 vs 172 ms; its 120 instructions per step cost more than the
				// round trips it saves while most buckets of a batch lose one or two entries)
				const u32 h = LANE, meta = w.bmetaV;
				u32 size = w.bsizeV;
				u32 tidNext = (u32)L.rq[ qoff];
				for (u32 q=0; q<tot; ++q)
				{
					const u32 tid = tidNext;
					{ const u32 nx = qoff + q + 1u; tidNext = (u32)L.rq[ nx < (u32)FAST_RQCAP ? nx : (u32)FAST_RQCAP - 1u]; }	// (requested beside this step's reads; unused after the last step)
					const u32 pos = ldLink<SP>( L, w, P, tid >> 2, tid & 3u) & 0xFFFu;	// (an earlier removal of this replay may have moved it)
					const u32 last = size - 1u;
					const uint2 m = ldEnt<SP>( L, w, P, h, meta, last);
					if (!SP || pos != last)
					{
						// cpp:133-152: the bucket's last entry moves into the hole (LDS-only instance: also when the hole IS the last
						// entry -- it writes the entry onto itself and the link of a trigger that is gone; a branch costs more)
						stEnt<SP>( L, w, P, h, meta, pos, m.x, m.y);
						stLink<SP>( L, w, P, (m.y & 0xFFFFu) >> 2, m.y & 3u, (h << 12) | pos);
					}
					size = last;
					if (SP) WAVE_FENCE();
				}
				L.bsize[ h] = size;
				w.bsizeV = size;
			}
			WAVE_FENCE();
			PROF_D( 11);
			w.nTrig -= total;
#ifdef SPA_PROF
			w.prof[ 8] += (u32)__builtin_amdgcn_readlane( waveScanMax( tot), 63); w.prof[ 9] += 1; w.prof[ 10] += nb;
#endif
		}
		if (nE)
		{
			// disposeRule (cpp:704-708): the ids of the row's rules are free again
			const bool fr = have && isExp;
			const u64 mL = __ballot( fr && r < (u32)R);
			if (fr && r < (u32)R) L.freeS[ w.freeN + (u32)__popcll( mL & lanesBelow())] = (typename FreeId<(R <= 256)>::type)r;
			w.freeN += (u32)__popcll( mL);
			if (SP)
			{
				const u64 mS = __ballot( fr && r >= (u32)R);
				if (fr && r >= (u32)R) w.sp[ P.spill.oFree + w.sFreeN + (u32)__popcll( mS & lanesBelow())] = r - R;
				w.sFreeN += (u32)__popcll( mS);
			}
		}
		WAVE_FENCE();
	}
}

// ---------------------------------------------------------------- expiry (cpp:1084-1135, window part: ranges are <= 63)
static __device__ __forceinline__ void setCurrentPos( LR L, Wave& w, KP P, u32 pos)
{
	// Called at the top of every event.  The dispose list of the transition before (w.nDispose rules, cpp:1030-1034) is still to be
	// deactivated: it joins the first expiry row that holds rules -- one deactivation batch per event --, or goes alone when the
	// position does not advance, no row holds rules, or the two together do not fit a batch.  (One call site: eight inlined copies of
	// the deactivation were a quarter of the kernel's code.)
	const u32 W = 1u << P.expShift;			// every live rule expires within the next W positions: one row per position
	u32 wcnt = 0;
	while (!w.err)
	{
		// the next position with rules to expire, if the position advances that far
		u32 n = 0, row = 0;
		while (wcnt < W && w.curpos < pos && !n)
		{
			row = w.curpos & (W-1u);
			n = (u32)__builtin_amdgcn_readlane( w.expCntV, row);
			++wcnt; ++w.curpos;
		}
		u32 nD = w.nDispose;
		if (!n && !nD) break;
		if (nD && n && nD + n > 64u) { n = 0; --wcnt; --w.curpos; }		// (the dispose list alone first; the row comes round again)
		w.nDispose = 0;
		// the rules of the row go last defined first (the reference's list is LIFO)
		if (w.spill) deactivateList<true>( L, w, P, nD, n, row); else deactivateList<false>( L, w, P, nD, n, row);
		if (n)
		{
			if (LANE == 0) L.expCnt[ row] = 0;
			if (LANE == row) w.expCntV = 0;
			WAVE_FENCE();
		}
	}
	if (w.curpos < pos) w.curpos = pos;
	if (w.spill)
	{
		// back to the LDS-only instance once nothing of the document is in the spill area any more
		bool over = false;
		if (LANE < 16u) over = w.bsizeV > (w.bmetaV >> 16);
		if (w.expCntV > ((u32)FAST_EXPCAP >> P.expShift)) over = true;
		if (!__ballot( over) && w.sFreeN == w.usedS) { w.spill = 0; w.sFreeN = 0; w.usedS = 0; }
	}
}

// ---------------------------------------------------------------- installEventPrograms (cpp:1137-1157), 64 programs at a time
// Lane l instantiates program l of the key event's list.  Everything whose ORDER is observable lands where
// the sequential loop would have put it: bucket positions, expiry lists, results and dispose entries all
// advance in lane order through ballots and prefix sums.
template <bool SP>
static __device__ __forceinline__ void installBatchT( LR L, Wave& w, KP P, u32 kb, u32 nb, u32 sord, u64 matMaskIn, u32 rIn,
	const uint4 q0, const uint4 q1, const uint4 q2, const Sim sim)
{
	const u32 value = sim.value, count = sim.count, end = sim.end, startLex = sim.startLex, nItems = sim.nItems, it0 = sim.it0, it1 = sim.it1, it2 = sim.it2;
	const bool hasStart = (sim.flags & S_HASSTART) != 0, done = (sim.flags & S_DONE) != 0, fin = (sim.flags & S_FIN) != 0, del = (sim.flags & S_DEL) != 0, resultNow = (sim.flags & S_RESULT) != 0;
	const bool have = LANE < nb;
	const u32 handle = q0.x, fmt = q0.y, meta = q0.w;
	const u32 tEv[ 3] = {q1.x, q1.z, q2.x};
	const u32 tInfo[ 3] = {q1.y, q1.w, q2.y};
	const u32 range = (meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK;
	const u64 matMask = matMaskIn;
	const bool mat = (matMask >> LANE) & 1ull;
	const u32 nmat = (u32)__popcll( matMask);
	const u32 r = rIn;
	// ---- expiry row of position sord+range (cpp:1066-1082): appended in lane order (= definition order)
	if (nmat)
	{
		const u32 row = (sord + range) & ((1u << P.expShift) - 1u);
		u64 same = matMask;
#pragma unroll
		for (int k=0; k<6; ++k)
		{
			const u64 bk = __ballot( mat && ((row >> k) & 1u));
			same &= ((row >> k) & 1u) ? bk : ~bk;
		}
		if (mat)
		{
			const u32 cnt = (u32)L.expCnt[ row];
			stExp<SP>( L, w, P, row, cnt + (u32)__popcll( same & lanesBelow()), r);
			WAVE_FENCE();
			if (!(same >> LANE >> 1)) L.expCnt[ row] = (u16)(cnt + (u32)__popcll( same));	// the last lane of a position closes its group
		}
	}
	// ---- triggers: bucket positions in (program, template) order (cpp:1204-1250 -> EventTriggerTable::add :114-131)
	u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
	u32 hB[ 3]; bool inst[ 3];
#pragma unroll
	for (int j=0; j<3; ++j)
	{
		hB[ j] = (tInfo[ j] >> FTI_BUCKET_SHIFT) & 15u;
		inst[ j] = mat && (tInfo[ j] & FTI_INSTALL);
		if (inst[ j]) byteInc( c0, c1, c2, c3, hB[ j]);
	}
	u32 tmask = 0;
	if (__ballot( inst[ 0] || inst[ 1] || inst[ 2]))
	{
		u32 i0 = waveScanAdd( c0), i1 = waveScanAdd( c1), i2 = waveScanAdd( c2), i3 = waveScanAdd( c3);	// fields stay < 256 (64 x 3)
		const u32 e0 = i0 - c0, e1 = i1 - c1, e2 = i2 - c2, e3 = i3 - c3;
		const u32 t0 = (u32)__builtin_amdgcn_readlane( i0, 63), t1 = (u32)__builtin_amdgcn_readlane( i1, 63);
		const u32 t2 = (u32)__builtin_amdgcn_readlane( i2, 63), t3 = (u32)__builtin_amdgcn_readlane( i3, 63);
		u32 oldSize = 0, add = 0;
		if (LANE < 16u) { oldSize = L.bsize[ LANE]; add = byteField( t0, t1, t2, t3, LANE); }
#pragma unroll
		for (int j=0; j<3; ++j)
		{
			if (inst[ j])
			{
				const u32 h = hB[ j];
				u32 sameB = 0;
#pragma unroll
				for (int jj=0; jj<j; ++jj) if (inst[ jj] && hB[ jj] == h) ++sameB;	// my own earlier templates
				const u32 pos = L.bsize[ h] + byteField( e0, e1, e2, e3, h) + sameB;
				const u32 sv = (tInfo[ j] & FTI_SIGVAL_MASK) | (((tInfo[ j] >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) << 4) | ((tInfo[ j] & FTI_HASVAR) ? 0x80u : 0u);
				stEnt<SP>( L, w, P, h, L.bmeta[ h], pos, tEv[ j], (4*r + (u32)j) | (sv << 16) | ((tInfo[ j] >> FTI_VAR_SHIFT) << 24));
				stLink<SP>( L, w, P, r, (u32)j, (h << 12) | pos);
				tmask |= 1u << j;
			}
		}
		WAVE_FENCE();
		if (LANE < 16u && add) L.bsize[ LANE] = oldSize + add;
		{
			const u32 s = t0 + t1 + t2 + t3;
			w.nTrig += (s & 0xFFu) + ((s >> 8) & 0xFFu) + ((s >> 16) & 0xFFu) + (s >> 24);	// (all fields together count <= 64 x 3 installs: no carries)
		}
	}
	// ---- the rule itself
	if (mat)
	{
		u32 hw = value | (count << H_COUNT_SHIFT) | H_ACTIVE | H_COLD | (tmask << H_TMASK_SHIFT) | (nItems << H_NITEMS_SHIFT);	// (dynamic batch: start and items in the rule's record)
		hw |= end ? ((end & 0xFFu) << H_END_SHIFT) : (u32)H_ENDZERO;
		if (done) hw |= H_DONE;
		if (hasStart) hw |= H_HASSTART;
		if (del || fin) hw |= H_LISTED;
		stHot<SP>( L, w, P, r, hw);
		u32* cold = &w.sp[ P.spill.oCold + 8*r];
		st4( cold, handle, fmt, startLex, it0);
		*(uint2*)(cold + 4) = make_uint2( it1, it2);
	}
	// ---- results (cpp:954-965), in lane order; items latest first
	{
		const bool emit = have && resultNow && handle != 0;
		const u64 rm = __ballot( emit);
		if (rm)
		{
			const u32 nr = (u32)__popcll( rm);
			if (w.nStaged + nr > P.spill.maxStaged) { FALLBACK( FB_STAGED); return; }
			if (emit)
			{
				const u32 ia = nItems == 3 ? it2 : nItems == 2 ? it1 : it0;		// items latest first
				const u32 ib = nItems == 3 ? it1 : it0;
				const u32 ic = it0;
				const u32 vars = nItems == 0 ? 0u : nItems == 1 ? (ia >> 24) : nItems == 2 ? ((ia >> 24) | ((ib >> 24) << 8)) : ((ia >> 24) | ((ib >> 24) << 8) | ((ic >> 24) << 16));
				stageResult( w, P, w.nStaged + (u32)__popcll( rm & lanesBelow()), handle, fmt, startLex, w.lbase, nItems, vars, ia & 0xFFFFFFu, ib & 0xFFFFFFu, ic & 0xFFFFFFu);
			}
			w.nStaged += nr;
			u32 incl = waveScanAdd( emit ? nItems : 0u);
			w.nStagedItems += (u32)__builtin_amdgcn_readlane( incl, 63);
		}
	}
	// ---- rules that finished or were deleted by their own key event: deactivated after the installs (cpp:1030-1034)
	{
		const bool wantDispose = mat && (del || fin);
		const u64 dm = __ballot( wantDispose);
		if (dm)
		{
			const u32 nd = (u32)__popcll( dm);
			if (w.nDispose + nd > P.spill.maxRules) { FALLBACK( FB_DISPOSE); return; }
			if (w.nDispose + nd > (u32)FAST_LISTCAP) w.spill = 1;
			if (wantDispose)
			{
				const u32 at = w.nDispose + (u32)__popcll( dm & lanesBelow());
				if (at < (u32)FAST_LISTCAP) L.list[ at] = (u16)r; else w.sp[ P.spill.oList + at] = r;
			}
			w.nDispose += nd;
		}
	}
	WAVE_FENCE();
}

// The same for a STATIC batch (l2_fast.h): no program of it is alternative-keyed, so the host has fired the key
// triggers and counted every rank; the lanes place their records and the wave adds the batch's totals.
template <bool SP>
static __device__ __forceinline__ void installStaticT( LR L, Wave& w, KP P, u32 ki0, u32 nb, u32 sord, u32 r,
	const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3)
{
	const bool have = LANE < nb;
	const u32 handle = q0.x, fmt = q0.y, meta = q0.w;
	const u32 tEv[ 3] = {q1.x, q1.z, q2.x};
	const u32 tInfo[ 3] = {q1.y, q1.w, q2.y};
	const u32 hw0 = q2.z, ranksA = q2.w, ranksB = q3.x, fl = q3.w;
	const u32 totals = bcast0( q3.y), itemsTotal = bcast0( q3.z) >> 24;		// (the same in every line of the batch)
	const u32 range = (meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK;
	const u32 row = (sord + range) & ((1u << P.expShift) - 1u);
	// ---- expiry row of position sord+range (cpp:1066-1082) and the triggers (cpp:1204-1250 -> EventTriggerTable::add :114-131)
	u32 cnt = 0, oldB[ 3] = {0,0,0};
	if (have)
	{
		cnt = (u32)L.expCnt[ row];
		stExp<SP>( L, w, P, row, cnt + (ranksA & 0xFFu), r);
#pragma unroll
		for (int j=0; j<3; ++j)
		{
			if (tInfo[ j] & FTI_INSTALL)
			{
				const u32 h = (tInfo[ j] >> FTI_BUCKET_SHIFT) & 15u;
				oldB[ j] = L.bsize[ h];
				const u32 pos = oldB[ j] + ((ranksB >> (8*j)) & 0xFFu);
				const u32 sv = (tInfo[ j] & FTI_SIGVAL_MASK) | (((tInfo[ j] >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) << 4) | ((tInfo[ j] & FTI_HASVAR) ? 0x80u : 0u);
				stEnt<SP>( L, w, P, h, L.bmeta[ h], pos, tEv[ j], (4*r + (u32)j) | (sv << 16) | ((tInfo[ j] >> FTI_VAR_SHIFT) << 24));
				stLink<SP>( L, w, P, r, (u32)j, (h << 12) | pos);
			}
		}
	}
	WAVE_FENCE();
	const u32 nItems = P.withItems ? ((hw0 >> H_NITEMS_SHIFT) & H_NITEMS_MASK) : 0u;
	if (have)
	{
		// the last lane of an expiry group / of a bucket's entries closes it (every lane has read the old counts above)
		const u32 close = (ranksA >> 8) & 0xFFu;
		if (close) L.expCnt[ row] = (u16)(cnt + close);
#pragma unroll
		for (int j=0; j<3; ++j)
		{
			if ((tInfo[ j] & FTI_INSTALL) && ((ranksB >> (24+j)) & 1u)) L.bsize[ (tInfo[ j] >> FTI_BUCKET_SHIFT) & 15u] = oldB[ j] + ((ranksB >> (8*j)) & 0xFFu) + 1u;
		}
		// ---- the rule itself: its word, its install line, its key lexem
		u32 hw = (hw0 & ~(H_NITEMS_MASK << H_NITEMS_SHIFT)) | (nItems << H_NITEMS_SHIFT);
		hw |= (fl & FKF_END_SET) ? (((sord + 1u) & 0xFFu) << H_END_SHIFT) : (u32)H_ENDZERO;
		stHot<SP>( L, w, P, r, hw);
		stKl<SP>( L, w, P, r, ki0 + LANE, w.lbase);
	}
	// ---- results (cpp:954-965), in lane order; the items are the key lexem under the key triggers' variables, latest first
	const u32 nres = (totals >> 16) & 0xFFu;
	if (nres)
	{
		if (w.nStaged + nres > P.spill.maxStaged) { FALLBACK( FB_STAGED); return; }
		if (have && (fl & FKF_RESULT_NOW))
		{
			const u32 v0 = q3.z & 0xFFu, v1 = (q3.z >> 8) & 0xFFu, v2 = (q3.z >> 16) & 0xFFu;
			const u32 va = nItems == 3 ? v2 : nItems == 2 ? v1 : v0;
			const u32 vb = nItems == 3 ? v1 : v0;
			const u32 vars = nItems == 0 ? 0u : nItems == 1 ? va : nItems == 2 ? (va | (vb << 8)) : (va | (vb << 8) | (v0 << 16));
			stageResult( w, P, w.nStaged + ((ranksA >> 16) & 0xFFu), handle, fmt, (fl & FKF_START_SET) ? w.lbase : 0u, w.lbase, nItems, vars, w.lbase, w.lbase, w.lbase);
		}
		w.nStaged += nres;
		if (P.withItems) w.nStagedItems += itemsTotal;
	}
	// ---- rules that finished or were deleted by their own key event: deactivated after the installs (cpp:1030-1034)
	const u32 nd = totals >> 24;
	if (nd)
	{
		if (w.nDispose + nd > P.spill.maxRules) { FALLBACK( FB_DISPOSE); return; }
		if (w.nDispose + nd > (u32)FAST_LISTCAP) w.spill = 1;
		if (have && (fl & FKF_DISPOSE_NOW))
		{
			const u32 at = w.nDispose + (ranksA >> 24);
			if (at < (u32)FAST_LISTCAP) L.list[ at] = (u16)r; else w.sp[ P.spill.oList + at] = r;
		}
		w.nDispose += nd;
	}
	w.nTrig += totals & 0xFFu;
	WAVE_FENCE();
}

// ---- rule ids for the lanes of matMask: LDS free stack, LDS bump, spill free stack, spill bump.  false: the document is handed over
static __device__ __forceinline__ bool allocRuleIds( LR L, Wave& w, KP P, u64 matMask, u32& r)
{
	const bool mat = (matMask >> LANE) & 1ull;
	const u32 nmat = (u32)__popcll( matMask);
	const u32 rank = (u32)__popcll( matMask & lanesBelow());
	r = 0;
	if (nmat)
	{
		const u32 a = w.freeN < nmat ? w.freeN : nmat;
		const u32 roomL = (u32)R - w.usedL;
		const u32 b = (nmat - a) < roomL ? (nmat - a) : roomL;
		const u32 rest = nmat - a - b;
		if (rest)
		{
			const u32 c = w.sFreeN < rest ? w.sFreeN : rest;
			const u32 d = rest - c;
			if ((u32)R + w.usedS + d > P.spill.maxRules) { FALLBACK( FB_RULES); return false; }
			w.spill = 1;
			if (mat && rank >= a+b)
			{
				if (rank < a+b+c) r = (u32)R + w.sp[ P.spill.oFree + w.sFreeN - 1 - (rank - a - b)];
				else r = (u32)R + w.usedS + (rank - a - b - c);
			}
			w.sFreeN -= c; w.usedS += d;
		}
		if (mat)
		{
			if (rank < a) r = (u32)L.freeS[ w.freeN - 1 - rank];
			else if (rank < a+b) r = w.usedL + (rank - a);
		}
		w.freeN -= a; w.usedL += b;
	}
	return true;
}
// ---- a bucket that outgrows its LDS region switches the document to the two-place accessors; one that outgrows its spill row
// hands the document over (false).  `bound`: entries the batch adds to one bucket at most; exact counts only when that is too many.
static __device__ __forceinline__ bool checkBuckets( LR L, Wave& w, KP P, u32 bound, const bool inst[ 3], const u32 hB[ 3])
{
	bool over = false, fail = false;
	if (LANE < 16u)
	{
		const u32 sz = w.bsizeV, cap = w.bmetaV >> 16;		// (register mirrors: no LDS round trip on the way)
		over = sz + bound > cap;
		fail = sz + bound > cap + (u32)FAST_SPILL_BUCKET || sz + bound > 0xFFFu;
	}
	if (__ballot( over))
	{
		u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
		for (int j=0; j<3; ++j) if (inst[ j]) byteInc( c0, c1, c2, c3, hB[ j]);
		const u32 i0 = waveScanAdd( c0), i1 = waveScanAdd( c1), i2 = waveScanAdd( c2), i3 = waveScanAdd( c3);
		const u32 t0 = (u32)__builtin_amdgcn_readlane( i0, 63), t1 = (u32)__builtin_amdgcn_readlane( i1, 63);
		const u32 t2 = (u32)__builtin_amdgcn_readlane( i2, 63), t3 = (u32)__builtin_amdgcn_readlane( i3, 63);
		over = false; fail = false;
		if (LANE < 16u)
		{
			const u32 sz = w.bsizeV + byteField( t0, t1, t2, t3, LANE), cap = w.bmetaV >> 16;
			over = sz > cap;
			fail = sz > cap + (u32)FAST_SPILL_BUCKET || sz > 0xFFFu;
		}
		if (__ballot( fail)) { FALLBACK( FB_BUCKET_SIZE); return false; }
		if (__ballot( over)) w.spill = 1;
	}
	return true;
}

// ---- a COMPACT static batch (l2_fast.h): two 16-byte loads per lane (c0, c1: FastStatic), one round of LDS reads, stores.
template <bool SP>
static __device__ __forceinline__ void installCompactT( LR L, Wave& w, KP P, u32 ki0, u32 nb, u32 sord, u32 r,
	const uint4 c0, const uint4 c1, u32 row, u32 cnt, const u32 oldB[ 2], const u32 metaB[ 2])
{
	const bool have = LANE < nb;
	const u32 ev[ 2] = {c0.x, c0.z}, info[ 2] = {c0.y, c0.w};
	const u32 hw0 = c1.x, misc = c1.y, ranks = c1.z;
	const u32 totals = bcast0( c1.w), itemsTotal = (bcast0( c1.z) >> 16) & 0xFFu;		// (the same in every line of the batch)
	if (have)
	{
		// ---- expiry row of position sord+range (cpp:1066-1082), triggers (cpp:1204-1250 -> EventTriggerTable::add :114-131)
		stExp<SP>( L, w, P, row, cnt + ((misc >> FSM_EXPRANK_SHIFT) & 0xFFu), r);
#pragma unroll
		for (int t=0; t<2; ++t)
		{
			if (info[ t] & FSI_PRESENT)
			{
				const u32 h = (info[ t] >> FSI_BUCKET_SHIFT) & 15u, j = info[ t] >> FSI_SLOT_SHIFT;
				const u32 pos = oldB[ t] + ((info[ t] >> FSI_RANK_SHIFT) & 0xFFu);
				stEnt<SP>( L, w, P, h, metaB[ t], pos, ev[ t], (4*r + j) | ((info[ t] & 0xFFu) << 16) | (((info[ t] >> 8) & 0xFFu) << 24));
				stLink<SP>( L, w, P, r, j, (h << 12) | pos);
			}
		}
	}
	WAVE_FENCE();
	const u32 nItems = P.withItems ? ((hw0 >> H_NITEMS_SHIFT) & H_NITEMS_MASK) : 0u;
	if (have)
	{
		// the last lane of an expiry group / of a bucket's entries closes it (every lane has read the old counts before)
		const u32 close = (misc >> FSM_EXPCLOSE_SHIFT) & 0xFFu;
		if (close) L.expCnt[ row] = (u16)(cnt + close);
#pragma unroll
		for (int t=0; t<2; ++t) if (info[ t] & FSI_LAST) L.bsize[ (info[ t] >> FSI_BUCKET_SHIFT) & 15u] = oldB[ t] + ((info[ t] >> FSI_RANK_SHIFT) & 0xFFu) + 1u;
		// ---- the rule itself: its word, its install line + key lexem
		u32 hw = (hw0 & ~(H_NITEMS_MASK << H_NITEMS_SHIFT)) | (nItems << H_NITEMS_SHIFT);
		hw |= (misc & FSM_END_SET) ? (((sord + 1u) & 0xFFu) << H_END_SHIFT) : (u32)H_ENDZERO;
		stHot<SP>( L, w, P, r, hw);
		stKl<SP>( L, w, P, r, ki0 + LANE, w.lbase);
	}
	// ---- results (cpp:954-965), in lane order; the items are the key lexem under the key triggers' variables, latest first
	const u32 nres = (totals >> 16) & 0xFFu;
	if (nres)
	{
		if (w.nStaged + nres > P.spill.maxStaged) { FALLBACK( FB_STAGED); return; }
		if (have && (misc & FSM_RESULT_NOW))
			stageLineResult( w, P, w.nStaged + (ranks & 0xFFu), ki0 + LANE, (misc & FSM_START_SET) ? w.lbase : 0u, w.lbase, nItems, 0u, 0u, w.lbase);
		w.nStaged += nres;
		if (P.withItems) w.nStagedItems += itemsTotal;
	}
	// ---- rules that finished or were deleted by their own key event: deactivated after the installs (cpp:1030-1034)
	const u32 nd = totals >> 24;
	if (nd)
	{
		if (w.nDispose + nd > P.spill.maxRules) { FALLBACK( FB_DISPOSE); return; }
		if (w.nDispose + nd > (u32)FAST_LISTCAP) w.spill = 1;
		if (have && (misc & FSM_DISPOSE_NOW))
		{
			const u32 at = w.nDispose + ((ranks >> 8) & 0xFFu);
			if (at < (u32)FAST_LISTCAP) L.list[ at] = (u16)r; else w.sp[ P.spill.oList + at] = r;
		}
		w.nDispose += nd;
	}
	w.nTrig += totals & 0xFFu;
	w.nSignals += (totals >> 8) & 0xFFu;
	WAVE_FENCE();
}

// pc0, pc1: the compact lines of the event's first batch, requested before its bucket scan (the load is under way while the
// triggers fire)
static __device__ __forceinline__ void installBatch( LR L, Wave& w, KP P, u32 kb, u32 kc, u32 sord, const uint4 pc0, const uint4 pc1)
{
	for (u32 base=0; base<kc && !w.err; base+=64)
	{
		const u32 nb = (kc - base) < 64u ? (kc - base) : 64u;
		const bool have = LANE < nb;
		uint4 c0 = pc0, c1 = pc1;
		if (base)
		{
			c0 = make_uint4( 0,0,0,0); c1 = c0;
			if (have) { const u32* S = (const u32*)&P.statics[ kb + base + LANE]; c0 = ld4( S); c1 = ld4( S+4); }
		}
#ifdef SPA_PROF_INSTALL
		u64 pi0 = __builtin_amdgcn_s_memtime();
#define PROF_I( SLOT, V) do { if (__ballot( (V) == 0xFFFFFFF1u)) w.err = SPD_ERR_INTERNAL; const u64 pi1 = __builtin_amdgcn_s_memtime(); w.prof[ SLOT] += pi1 - pi0; pi0 = pi1; } while (0)
#else
#define PROF_I( SLOT, V) do {} while (0)
#endif
		if ((bcast0( c1.y) & FSM_BATCH_COMPACT) && sord != 0)
		{
			PROF_I( 6, c1.y);
			// ---- compact static batch: everything but the absolute positions is in the lines
			const u32 info[ 2] = {c0.y, c0.w};
			const u32 row = (sord + (c1.y & FSM_RANGE_MASK)) & ((1u << P.expShift) - 1u);
			u32 cnt = 0, oldB[ 2] = {0,0}, metaB[ 2] = {0,0};
			bool inst[ 3] = {false,false,false}; u32 hB[ 3] = {0,0,0};
			if (have)
			{
				cnt = (u32)L.expCnt[ row];
#pragma unroll
				for (int t=0; t<2; ++t)
				{
					if (info[ t] & FSI_PRESENT) { hB[ t] = (info[ t] >> FSI_BUCKET_SHIFT) & 15u; inst[ t] = true; oldB[ t] = L.bsize[ hB[ t]]; metaB[ t] = L.bmeta[ hB[ t]]; }
				}
			}
			w.nInstalled += nb;
			u32 r;
			if (!allocRuleIds( L, w, P, __ballot( have), r)) return;
			if (!w.spill)
			{
				if (__ballot( have && cnt + nb > ((u32)FAST_EXPCAP >> P.expShift))) w.spill = 1;	// (upper bound: the whole batch in my row)
			}
			if (!checkBuckets( L, w, P, bcast0( c1.w) & 0xFFu, inst, hB)) return;
			PROF_I( 7, cnt + oldB[ 0] + oldB[ 1] + r);
			if (w.spill) installCompactT<true>( L, w, P, kb + base, nb, sord, r, c0, c1, row, cnt, oldB, metaB);
			else installCompactT<false>( L, w, P, kb + base, nb, sord, r, c0, c1, row, cnt, oldB, metaB);
			PROF_I( 11, 0u);
			w.bsizeV = L.bsize[ LANE & 15u];
			continue;
		}
		uint4 q0 = make_uint4( 0,0,0,0), q1 = q0, q2 = q0, q3 = q0;
		if (have)
		{
			const FastKeyInst* K = &P.keyinst[ kb + base + LANE];
			q0 = ld4( K); q1 = ld4( (const u32*)K + 4); q2 = ld4( (const u32*)K + 8); q3 = ld4( (const u32*)K + 12);
		}
		// a static batch at a position other than 0: the key fires and all ranks are in the lines (l2_fast.h)
		const bool isStatic = (bcast0( q3.w) & FKF_BATCH_STATIC) != 0 && sord != 0;
		const u32 pastEvent = q0.z, meta = q0.w;
		const u32 tEv[ 3] = {q1.x, q1.z, q2.x};
		const u32 tInfo[ 3] = {q1.y, q1.w, q2.y};		// (templates beyond the program's count are all zero)
		const u32 range = (meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK;
		// ---- signals on the fresh slot, in registers: (1) an alternative-keyed program replays the logged
		//      original key event (cpp:1253-1258 -> :1272-1334), (2) the key trigger(s) fire (cpp:1259-1269)
		Sim sim;
		sim.value = meta & FKI_VALUE_MASK; sim.count = (meta >> FKI_COUNT_SHIFT) & FKI_COUNT_MASK; sim.end = 0;
		sim.startLex = 0; sim.nItems = 0; sim.it0 = 0; sim.it1 = 0; sim.it2 = 0; sim.nFires = 0; sim.flags = 0;
		bool dropped = false;		// deactivated by the replay: the rule never becomes visible to anything else
		if (!isStatic && __ballot( have && pastEvent != 0))
		{
			const u32 psi = (have && pastEvent) ? ((meta >> FKI_PASTSTOP_SHIFT) & FKI_PASTSTOP_MASK) : 0u;
			// the stop-word log is in registers, lane = stop word: every lane fetches (all lanes take part in a bpermute)
			const u32 plex = (u32)__builtin_amdgcn_ds_bpermute( (int)((psi-1u) << 2), (int)w.stopLex);
			const u32 psord = (u32)__builtin_amdgcn_ds_bpermute( (int)((psi-1u) << 2), (int)w.stopOrd);
			const u32 pts = (u32)__builtin_amdgcn_ds_bpermute( (int)((psi-1u) << 2), (int)w.stopTs);
			u32 delTs[ 3];
#pragma unroll
			for (int j=0; j<3; ++j)
			{
				const u32 esi = (tInfo[ j] >> FTI_DELSTOP_SHIFT) & FTI_DELSTOP_MASK;
				delTs[ j] = (u32)__builtin_amdgcn_ds_bpermute( (int)((esi-1u) << 2), (int)w.stopTs);
				if (!esi) delTs[ j] = 0;
			}
			if (psi)
			{
				if (pts && psord + range >= w.curpos)
				{
#pragma unroll
					for (int j=2; j>=0; --j)		// the rule's trigger list: last installed first
					{
						if ((tInfo[ j] & FTI_INSTALL) && tEv[ j] == pastEvent) fireLocal( sim, tInfo[ j], psord, plex, P.withItems);
					}
					// a structure delimiter logged after the replayed event cancels the rule (cpp:1306-1321)
					bool cancelled = false;
#pragma unroll
					for (int j=0; j<3; ++j)
					{
						if ((tInfo[ j] & FTI_INSTALL) && ((tInfo[ j] >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) == SIG_DEL)
						{
							const u32 ts = delTs[ j];
							if (ts && ts > pts) cancelled = true;
						}
					}
					if (cancelled || (sim.flags & (S_DEL | S_FIN))) dropped = true;	// (replayPastEvent deactivates what its signals disposed, cpp:1322-1330)
				}
			}
			w.nAlt += (u32)__popcll( __ballot( have && pastEvent != 0));
		}
		if (!isStatic)
		{
			if (have && !dropped)
			{
#pragma unroll
				for (int j=0; j<3; ++j) if (tInfo[ j] & FTI_KEY) fireLocal( sim, tInfo[ j], sord, w.lbase, P.withItems);
			}
			if (__ballot( (sim.flags & S_ODD) != 0)) { FALLBACK( FB_ITEMS); return; }
		}
		const bool mat = have && !dropped;
		const u64 matMask = __ballot( mat);
		const u32 nmat = (u32)__popcll( matMask);
		// ---- statistics (cpp:1251, :780)
		w.nInstalled += nb;
		if (isStatic) w.nSignals += (bcast0( q3.y) >> 8) & 0xFFu;
		else
		{
			u32 incl = waveScanAdd( have ? sim.nFires : 0u);
			w.nSignals += (u32)__builtin_amdgcn_readlane( incl, 63);
		}
		u32 r;
		if (!allocRuleIds( L, w, P, matMask, r)) return;
		// ---- an expiry row or a bucket that outgrows its LDS region switches the document to the two-place accessors
		if (nmat && !w.spill)
		{
			const u32 row = (sord + range) & ((1u << P.expShift) - 1u);
			const bool over = mat && (u32)L.expCnt[ row] + nmat > ((u32)FAST_EXPCAP >> P.expShift);	// (upper bound: the whole batch in my row)
			if (__ballot( over)) w.spill = 1;
		}
		{
			bool inst[ 3]; u32 hB[ 3];
#pragma unroll
			for (int j=0; j<3; ++j) { inst[ j] = mat && (tInfo[ j] & FTI_INSTALL); hB[ j] = (tInfo[ j] >> FTI_BUCKET_SHIFT) & 15u; }
			// upper bound without the scan: the batch adds at most nmat x 3 entries to a bucket (a static batch: its total)
			if (!checkBuckets( L, w, P, isStatic ? (bcast0( q3.y) & 0xFFu) : 3*nmat, inst, hB)) return;
		}
		if (isStatic)
		{
			if (w.spill) installStaticT<true>( L, w, P, kb + base, nb, sord, r, q0, q1, q2, q3);
			else installStaticT<false>( L, w, P, kb + base, nb, sord, r, q0, q1, q2, q3);
		}
		else if (w.spill) installBatchT<true>( L, w, P, kb, nb, sord, matMask, r, q0, q1, q2, sim);
		else installBatchT<false>( L, w, P, kb, nb, sord, matMask, r, q0, q1, q2, sim);
		w.bsizeV = L.bsize[ LANE & 15u];		// (the mirror the next batch's capacity check reads)
	}
	// the register mirrors of the bucket sizes and the expiry row lengths (requested now, needed at the next event at the earliest)
	w.bsizeV = L.bsize[ LANE & 15u];
	w.expCntV = (u32)L.expCnt[ LANE];
}

// ---------------------------------------------------------------- doTransition (cpp:981-1064) for an input term
template <bool SP>
static __device__ __forceinline__ void scanAndFire( LR L, Wave& w, KP P, u32 id, u32 ordpos)
{
	// fire the triggers waiting for this event: 64 bucket entries per step, one ballot
	const u32 h = evhash( id) & 15u;
	const u32 n = (u32)__builtin_amdgcn_readlane( w.bsizeV, h);
	const u32 meta = (u32)__builtin_amdgcn_readlane( w.bmetaV, h);
	for (u32 base=0; base<n && !w.err; base+=64)
	{
		const u32 i = base + LANE;
		uint2 en = make_uint2( 0, 0);
		if (i < n) en = ldEnt<SP>( L, w, P, h, meta, i);		// (whole entries: a hit's trigger word comes out of the lane's register, not out of a second read)
		u64 m = __ballot( i < n && en.x == id);
		if (!SP && (m & (m - 1ull)) != 0)
		{
			// several hits: all at once, a lane each
			if (fireBatch( L, w, P, m, en.y, ordpos)) m = 0;
		}
		while (m && !w.err)
		{
			const u32 p = (u32)__builtin_ctzll( m);
			m &= m-1;
			const u32 tsv = (u32)__builtin_amdgcn_readlane( en.y, p);
			fireSignal<SP>( L, w, P, tsv, ordpos);
		}
	}
}

static __device__ __forceinline__ void runKernel()
{
	KP P = kernelParams();
	__shared__ Lds ldsDoc;
	LR L = *(__attribute__((address_space(3))) Lds*)&ldsDoc;
	Wave w;
	w.sp = P.spillBase + (u64)blockIdx.x * P.spill.totalWords;
	const u32 ndocs = P.ndocs;
	const u32 waveSlot = blockIdx.x, nWaveSlots = gridDim.x;

	for (u32 round=0; round<=ndocs; ++round)
	{
		u32 doc = waveSlot;
		if (round)
		{
			u32 nx = 0;
			if (LANE == 0) nx = atomicAdd( P.docCursor, 1u);
			doc = nWaveSlots + bcast0( nx);
		}
		if (doc >= ndocs) break;
		// per-document reset
		w.bmetaV = P.bucketMeta[ LANE & 15u]; w.bsizeV = 0; w.expCntV = 0;
		if (LANE < 16u) { L.bsize[ LANE] = 0; L.bmeta[ LANE] = w.bmetaV; }
		L.expCnt[ LANE] = 0;
		w.stopLex = 0; w.stopOrd = 0; w.stopTs = 0; w.posLexems = 0;
		w.curpos = 0; w.timestamp = 0; w.nInstalled = 0; w.nAlt = 0; w.nSignals = 0; w.nTrig = 0; w.openLo = 0; w.openHi = 0;
		w.freeN = 0; w.usedL = 0; w.sFreeN = 0; w.usedS = 0;
		w.nDispose = 0; w.nStaged = 0; w.nStagedItems = 0; w.err = 0; w.lbase = 0; w.why = 0; w.spill = 0;
#ifdef SPA_PROF
		for (int pi=0; pi<12; ++pi) w.prof[ pi] = 0;
#endif
		WAVE_FENCE();

		u64 lbeg, lend;
		if (P.docRangesIn)
		{
			const u32* rp = (const u32*)&P.docRangesIn[ 2*(u64)doc];
			lbeg = ((u64)ldu( rp+1) << 32) | ldu( rp);
			lend = lbeg + (((u64)ldu( rp+3) << 32) | ldu( rp+2));
		}
		else
		{
			lbeg = ((u64)ldu( (const u32*)&P.docOffsets[ doc]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc]);
			lend = ((u64)ldu( (const u32*)&P.docOffsets[ doc+1]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc+1]);
		}
		if (lend - lbeg >= (1ull << 24)) FALLBACK( FB_LEXEMS);	// lexem indices are kept in 24 bits beside the variable
		u32 curPosition = 0, nEvents = 0;
		for (u64 tile=lbeg; tile<lend && !w.err; tile+=64)
		{
			// coalesced fetch of up to 64 lexems (16 B each); every lane also looks its own lexem's event up in the key
			// table, so the table's latency is paid once per 64 events
			PROF_DECL;
			uint4 lx = make_uint4( 0,0,0,0); u32 seg = 0;
			const bool mine = tile + LANE < lend;
			if (mine)
			{
				lx = ((const uint4*)P.lexems)[ tile + LANE];
				if (P.origseg) seg = P.origseg[ tile + LANE];
			}
			u32 kBegin = 0, kCount = 0, kStop = 0;
			if (mine && lx.x && lx.x < (1u<<29))
			{
				u32 slot = keyHash( lx.x) & P.keymask;
				for (u32 probes=0; probes<=P.keymask; ++probes)
				{
					const uint4 eq = ld4( &P.keytab[ slot]);		// {event, kiBegin, kiCount, stopIdx}
					if (eq.x == lx.x) { kBegin = eq.y; kCount = eq.z; kStop = eq.w; break; }
					if (eq.x == 0) break;
					slot = (slot+1) & P.keymask;
				}
			}
			const u32 cnt = (lend - tile) < 64 ? (u32)(lend - tile) : 64u;
#ifdef SPA_PROF
			if (__ballot( kBegin == 0xFFFFFFFFu)) break;	// (the probes complete here)
#endif
			PROF_ADD( 5);
			for (u32 k=0; k<cnt && !w.err; ++k)
			{
				const u32 id = __builtin_amdgcn_readlane( lx.x, k), ordpos = __builtin_amdgcn_readlane( lx.y, k);
				const u32 origpos = __builtin_amdgcn_readlane( lx.z, k), origsize = __builtin_amdgcn_readlane( lx.w, k);
				const u32 origseg = __builtin_amdgcn_readlane( seg, k);
				// PatternMatcherContext::putInput (patternMatcher.cpp:131-162)
				if (curPosition > ordpos) { w.err = SPD_ERR_ORDER; break; }
				else if (curPosition < ordpos) { curPosition = ordpos; w.posLexems = 0; }
				else if (origsize >= 0x7FFFFFFFu || origseg >= 0x7FFFFFFFu || origpos >= 0x7FFFFFFFu) { w.err = SPD_ERR_RANGE; break; }
				// expiry up to the event's position + the rules the transition before has finished or deleted
				if (w.curpos != ordpos || w.nDispose) { setCurrentPos( L, w, P, ordpos); PROF_ADD( 3); if (w.err) break; }
				if (id >= (1u<<29)) { w.err = SPD_ERR_RANGE; break; }
				w.lbase = (u32)(tile - lbeg) + k;
				if (++w.posLexems > 60u) { FALLBACK( FB_POSLEXEMS); break; }	// (key lexems are kept modulo 4096: ldKl)
				// ---- doTransition (cpp:981-1064) for an input term: no follow events in a flat rule set
				{
					const u32 lo = w.openLo + w.nTrig;
					if (lo < w.openLo) w.openHi += 1;
					w.openLo = lo;
				}
				// the programs keyed by this event: their (compact) install lines are requested now and read after the bucket scan
				const u32 kb = __builtin_amdgcn_readlane( kBegin, k), kc = __builtin_amdgcn_readlane( kCount, k);
				const u32 stopIdx = __builtin_amdgcn_readlane( kStop, k);
				uint4 pc0 = make_uint4( 0,0,0,0), pc1 = pc0;
				if (LANE < kc) { const u32* S = (const u32*)&P.statics[ kb + LANE]; pc0 = ld4( S); pc1 = ld4( S+4); }
				if (id)
				{
					if (w.spill) scanAndFire<true>( L, w, P, id, ordpos); else scanAndFire<false>( L, w, P, id, ordpos);
				}
				PROF_ADD( 0);
				if (w.err) break;
				if (kc) installBatch( L, w, P, kb, kc, ordpos, pc0, pc1);
				PROF_ADD( 1);
				if (w.err) break;
				// (the rules that finished or were deleted are deactivated at the top of the next event, together with the rules that expire there)
				WAVE_FENCE();
				if (stopIdx)
				{
					if (LANE == stopIdx-1u) { w.stopLex = w.lbase; w.stopOrd = ordpos; w.stopTs = w.timestamp + 1u; }
					w.timestamp += 1;
					WAVE_FENCE();
				}
				PROF_ADD( 2);
				++nEvents;
			}
		}

		// ---- fetchResults (patternMatcher.cpp:271-301): the staged results become sp_result_t / sp_result_item_t records
		PROF_DECL;
		u32 nres = w.err ? 0 : w.nStaged;
		const u32 nitems = (w.err || !P.withItems) ? 0 : w.nStagedItems;
		u64 resBase = 0, itemBase = 0;
		if (nres)
		{
			u64 b = 0;
			if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_RESULTS], (unsigned long long)nres);
			resBase = ((u64)bcast0( (u32)(b >> 32)) << 32) | bcast0( (u32)b);
			if (resBase + nres > P.resultCapacity) { w.err = SPD_ERR_OUTPUT; nres = 0; }
		}
		if (nres && nitems)
		{
			u64 b = 0;
			if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_ITEMS], (unsigned long long)nitems);
			itemBase = ((u64)bcast0( (u32)(b >> 32)) << 32) | bcast0( (u32)b);
			if (itemBase + nitems > P.itemCapacity) { w.err = SPD_ERR_OUTPUT; nres = 0; }
		}
		if (nres)
		{
			WAVE_FENCE();
			u64 ip = itemBase;
			for (u32 base=0; base<nres; base+=64)
			{
				const u32 ri = base + LANE;
				const bool hv = ri < nres;
				uint4 s0 = make_uint4( 0,0,0,0), s1 = s0;
				if (hv) { const u32* S = &w.sp[ P.spill.oStaged + 8*ri]; s0 = ld4( S); s1 = ld4( S+4); }
				if (hv && (s0.x & (u32)STAGED_LINE))
				{
					// handle, format and the variables of the static items come from the install line (latest item first)
					const u32* K = (const u32*)&P.keyinst[ s0.x & 0xFFFFFu];
					const u32 handle = K[ 0], fmt = K[ 1], vw = K[ 14];
					const u32 nStatic = s1.x & 3u, hasNew = (s1.x >> 2) & 1u, newVar = (s1.x >> 8) & 0xFFu, lx = s1.y, endLex = s0.w;
					u32 vv[ 3] = {0,0,0}, ll[ 3] = {0,0,0};
					u32 n = 0;
					if (hasNew) { vv[ 0] = newVar; ll[ 0] = endLex; n = 1; }
#pragma unroll
					for (int k=2; k>=0; --k)
					{
						if ((u32)k < nStatic)
						{
							const u32 v = (vw >> (8*k)) & 0xFFu;
							if (n == 0) { vv[ 0] = v; ll[ 0] = lx; } else if (n == 1) { vv[ 1] = v; ll[ 1] = lx; } else if (n == 2) { vv[ 2] = v; ll[ 2] = lx; }
							++n;
						}
					}
					s0.x = handle; s0.y = fmt;
					s1.x = n | (vv[ 0] << 8) | (vv[ 1] << 16) | (vv[ 2] << 24); s1.y = ll[ 0]; s1.z = ll[ 1]; s1.w = ll[ 2];
				}
				const u32 ni = P.withItems ? (s1.x & 0xFFu) : 0u;
				u32 incl = waveScanAdd( ni);
				const u64 mine = ip + (incl - ni);
				if (hv)
				{
					const uint4 a = ((const uint4*)P.lexems)[ lbeg + s0.z], z = ((const uint4*)P.lexems)[ lbeg + s0.w];	// first taken / matching lexem {id, ordpos, origpos, origsize}
					const u32 sa = P.origseg ? P.origseg[ lbeg + s0.z] : 0u, sz = P.origseg ? P.origseg[ lbeg + s0.w] : 0u;
					u32* o = P.results + (resBase + ri)*9;
					o[0] = s0.x; o[1] = a.y; o[2] = z.y + 1u; o[3] = sa; o[4] = a.z; o[5] = sz; o[6] = z.z + z.w;
					o[7] = P.withItems ? (u32)mine : 0u; o[8] = ni;
					if (P.withFormats) P.resultFormat[ resBase + ri] = s0.y;
					const u32 il[ 3] = {s1.y, s1.z, s1.w};
#pragma unroll
					for (int q=0; q<3; ++q)
					{
						if ((u32)q < ni)
						{
							const uint4 t = ((const uint4*)P.lexems)[ lbeg + il[ q]];
							const u32 st = P.origseg ? P.origseg[ lbeg + il[ q]] : 0u;
							u32* io = P.items + (mine + (u32)q)*7;
							io[0] = (s1.x >> (8 + 8*q)) & 0xFFu; io[1] = t.y; io[2] = t.y + 1u; io[3] = st; io[4] = t.z; io[5] = st; io[6] = t.z + t.w;
							if (P.withFormats) { P.itemFormat[ 2*(mine + (u32)q)] = 0; P.itemFormat[ 2*(mine + (u32)q)+1] = 0; }
						}
					}
				}
				ip += (u32)__builtin_amdgcn_readlane( incl, 63);
			}
		}
		PROF_ADD( 4);
#ifdef SPA_PROF
		if (LANE == 0 && P.prof) for (int pi=0; pi<12; ++pi) atomicAdd( (unsigned long long*)&P.prof[ pi], (unsigned long long)w.prof[ pi]);
#endif
		if (LANE == 0)
		{
			if (w.err == SPD_FAST_FALLBACK)
			{
				const u32 at = atomicAdd( P.fallbackCount, 1u);
				P.fallbackList[ at] = doc;
				P.docRange[ 2*(u64)doc] = 0; P.docRange[ 2*(u64)doc+1] = 0;
				P.docStatus[ doc] = (int32_t)SPD_FAST_FALLBACK;
				atomicAdd( (unsigned long long*)&P.counters[ SPC_HANDOVER], 1ull);
				if (P.diag) { atomicAdd( &P.diag[ 0], 1u); atomicAdd( &P.diag[ w.why & 15u], 1u); }
			}
			else
			{
				P.docRange[ 2*(u64)doc] = resBase; P.docRange[ 2*(u64)doc+1] = nres;
				u64* st = P.docStats + 4*(u64)doc;
				st[0] = w.nInstalled; st[1] = w.nAlt; st[2] = w.nSignals; st[3] = ((u64)w.openHi << 32) | w.openLo;
				P.docStatus[ doc] = (int32_t)w.err;
				atomicAdd( (unsigned long long*)&P.counters[ SPC_EVENTS], (unsigned long long)nEvents);
				if (w.err) atomicAdd( (unsigned long long*)&P.counters[ SPC_FAILED], 1ull);
			}
		}
	}
}
}; // Engine

} // anonymous namespace

// ================================================================== kernel instances (LDS capacities: rules, bucket entries)
// 4 waves per SIMD = 16 per CU is what 10 KB of LDS per document allow; the register budget that goes with it (128) is stated
// because the kernel is latency bound: one register more would cost a quarter of the documents in flight
#ifndef SPA_L2_FAST_WAVES_PER_EU
#define SPA_L2_FAST_WAVES_PER_EU 4
#endif
#define SPA_L2_FAST_OCC __attribute__((amdgpu_waves_per_eu( SPA_L2_FAST_WAVES_PER_EU, SPA_L2_FAST_WAVES_PER_EU)))
#define SPA_FAST_INSTANCE( NAME, RR, TT) \
	extern "C" __global__ __launch_bounds__(64) SPA_L2_FAST_OCC void NAME( FastParams kernelArgs) { Engine<RR,TT>::runKernel(); }
SPA_FAST_INSTANCE( spa_l2_fast_kernel_s, 192, 312)
SPA_FAST_INSTANCE( spa_l2_fast_kernel_m, 320, 512)
SPA_FAST_INSTANCE( spa_l2_fast_kernel_l, 512, 1024)
SPA_FAST_INSTANCE( spa_l2_fast_kernel_n, 256, 448)		// 10 KB of LDS: 16 waves per CU
SPA_FAST_INSTANCE( spa_l2_fast_kernel_t, 8, 128)		// tests: everything beyond a handful of rules runs through the spill area

namespace spa {
static const void* fastInstance( unsigned variant)
{
	return variant == 0 ? (const void*)spa_l2_fast_kernel_s : variant == 1 ? (const void*)spa_l2_fast_kernel_m : variant == 2 ? (const void*)spa_l2_fast_kernel_l : variant == 4 ? (const void*)spa_l2_fast_kernel_n : (const void*)spa_l2_fast_kernel_t;
}
void fastCapacities( unsigned variant, uint32_t& R, uint32_t& T)
{
	if (variant == 0) { R = 192; T = 312; } else if (variant == 1) { R = 320; T = 512; } else if (variant == 2) { R = 512; T = 1024; } else if (variant == 4) { R = 256; T = 448; } else { R = 8; T = 128; }
}
// resident single-wave workgroups per CU of a kernel instance (registers and LDS both limit it)
int fastBlocksPerCU( unsigned variant)
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor( &n, fastInstance( variant), 64, 0) != hipSuccess || n < 1) n = 1;
	if (n > 32) n = 32;
	return n;
}
hipError_t launchL2Fast( const FastParams& P, unsigned variant, unsigned nblocks, hipStream_t stream)
{
	if (variant == 0) hipLaunchKernelGGL( spa_l2_fast_kernel_s, dim3( nblocks), dim3( 64), 0, stream, P);
	else if (variant == 1) hipLaunchKernelGGL( spa_l2_fast_kernel_m, dim3( nblocks), dim3( 64), 0, stream, P);
	else if (variant == 2) hipLaunchKernelGGL( spa_l2_fast_kernel_l, dim3( nblocks), dim3( 64), 0, stream, P);
	else if (variant == 4) hipLaunchKernelGGL( spa_l2_fast_kernel_n, dim3( nblocks), dim3( 64), 0, stream, P);
	else hipLaunchKernelGGL( spa_l2_fast_kernel_t, dim3( nblocks), dim3( 64), 0, stream, P);
	return hipGetLastError();
}
}
