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
//         its FastKeyInst index, the bucket position of each of its <= 3 installed triggers (u16), its link in
//         the expiry list of its position (u16); the 16 trigger buckets of the reference's EventTriggerTable
//         (cpp:114-257) as {event, trigger id | signal byte} entries in 16-entry chunks from a shared pool, in
//         exactly the reference's positions (append at the end, swap-with-last removal); the 64 expiry list
//         heads; the stop-word log {lexem index, ordpos, timestamp}; the dispose list; ~20 scalars.
//   HBM   per rule instance ONE write-once 16-byte record {first taken lexem, <= 3 captured items as lexem
//         indices}, read only if the rule matches after its installation; staged results (32 B) expanded to
//         sp_result_t / sp_result_item_t records at the document's end; a spill area with the same record
//         shapes for rule ids / bucket chunks beyond the LDS capacities (bursts).
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
#define LDSQ __attribute__((address_space(3)))
typedef LDSQ u32 lu32;
typedef LDSQ u16 lu16;
typedef LDSQ u8 lu8;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(4))) FastParams& KP;
__device__ __forceinline__ KP kernelParams() { return *(const __attribute__((address_space(4))) FastParams*)__builtin_amdgcn_kernarg_segment_ptr(); }

// cross-lane hand-over through LDS / the wave's own spill area: LDS and vector memory operations of one
// wave execute in order; the fence keeps the compiler from moving or reusing accesses across the point
#define WAVE_FENCE() __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront")

// rule word
enum {
	H_VALUE_MASK=0xFu, H_COUNT_SHIFT=4, H_COUNT_MASK=0x1Fu, H_END_SHIFT=9, H_END_MASK=0xFFu, H_ENDZERO=1u<<17,
	H_DONE=1u<<18, H_ACTIVE=1u<<19, H_TMASK_SHIFT=20, H_TMASK_MASK=0x7u, H_NITEMS_SHIFT=23, H_NITEMS_MASK=0x3u, H_HASSTART=1u<<25
};
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
__device__ __forceinline__ u32 ldu( const lu32* p) { return __builtin_amdgcn_readfirstlane( *p); }
__device__ __forceinline__ u32 ldu( const lu16* p) { return __builtin_amdgcn_readfirstlane( (u32)*p); }
__device__ __forceinline__ u32 ldu( const lu8* p) { return __builtin_amdgcn_readfirstlane( (u32)*p); }
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
	const u32 inc = 1u << ((h & 3u)*8), ws = h >> 2;
	if (ws == 0) c0 += inc; else if (ws == 1) c1 += inc; else if (ws == 2) c2 += inc; else c3 += inc;
}

// ---------------------------------------------------------------- the wave's view of its state
struct Wave
{
	// LDS arrays
	lu32* sc; lu32* hot; lu16* link; lu16* next; lu16* freeS; lu32* ev; lu32* ts;
	lu8* ctab; lu8* cfree; lu32* bsize; lu32* bchunks; lu16* win; lu32* stop; lu16* list;
	u32* sp;		// spill + cold area in HBM
	u32 R, T;		// LDS capacities (rules, bucket entries)
	// per-document scalars (wave-uniform registers)
	u32 curpos, timestamp, nInstalled, nAlt, nSignals, nTrig, openLo, openHi;
	u32 freeN, usedL, sFreeN, usedS;	// rule ids: LDS free stack / bump, spill free stack / bump
	u32 cFreeN, cUsed;			// bucket chunks: free stack / bump
	u32 nDispose, nStaged, nStagedItems, err;
	u32 lbase;				// lexem index of the event being processed, relative to the document
	u32 why;				// reason of a hand-over (diagnostics)
#ifdef SPA_PROF
	u64 prof[ 12];
#endif
};

// ---- rule fields: ids < R in LDS, others in the spill area (same shapes)
__device__ __forceinline__ u32 ldHot( const Wave& w, KP P, u32 r) { return r < w.R ? w.hot[ r] : w.sp[ P.spill.oHot + (r - w.R)]; }
__device__ __forceinline__ void stHot( const Wave& w, KP P, u32 r, u32 v) { if (r < w.R) w.hot[ r] = v; else w.sp[ P.spill.oHot + (r - w.R)] = v; }
__device__ __forceinline__ u32 ldLink( const Wave& w, KP P, u32 r, u32 j) { return r < w.R ? (u32)w.link[ 3*r + j] : w.sp[ P.spill.oLink + 3*(r - w.R) + j]; }
__device__ __forceinline__ void stLink( const Wave& w, KP P, u32 r, u32 j, u32 v) { if (r < w.R) w.link[ 3*r + j] = (u16)v; else w.sp[ P.spill.oLink + 3*(r - w.R) + j] = v; }
__device__ __forceinline__ u32 ldNext( const Wave& w, KP P, u32 r) { return r < w.R ? (u32)w.next[ r] : w.sp[ P.spill.oNext + (r - w.R)]; }
__device__ __forceinline__ void stNext( const Wave& w, KP P, u32 r, u32 v) { if (r < w.R) w.next[ r] = (u16)v; else w.sp[ P.spill.oNext + (r - w.R)] = v; }
// ---- bucket entries: entry index = chunk id * 16 + offset; chunks < T/16 in LDS
__device__ __forceinline__ u32 entryIndex( const Wave& w, u32 h, u32 pos) { return (u32)w.ctab[ h*FAST_BUCKET_CHUNKS + (pos >> 4)]*FAST_CHUNK + (pos & 15u); }
__device__ __forceinline__ u32 ldEv( const Wave& w, KP P, u32 idx) { return idx < w.T ? w.ev[ idx] : w.sp[ P.spill.oEnt + 2*(idx - w.T)]; }
__device__ __forceinline__ u32 ldTs( const Wave& w, KP P, u32 idx) { return idx < w.T ? w.ts[ idx] : w.sp[ P.spill.oEnt + 2*(idx - w.T) + 1]; }
__device__ __forceinline__ void stEnt( const Wave& w, KP P, u32 idx, u32 e, u32 t)
{
	if (idx < w.T) { w.ev[ idx] = e; w.ts[ idx] = t; }
	else { *(uint2*)&w.sp[ P.spill.oEnt + 2*(idx - w.T)] = make_uint2( e, t); }
}
// ---- the list of rules to deactivate: first FAST_LISTCAP entries in LDS, the rest in the spill area
__device__ __forceinline__ u32 ldList( const Wave& w, KP P, u32 i) { return i < (u32)FAST_LISTCAP ? (u32)w.list[ i] : w.sp[ P.spill.oList + i]; }
__device__ __forceinline__ void stList( const Wave& w, KP P, u32 i, u32 r) { if (i < (u32)FAST_LISTCAP) w.list[ i] = (u16)r; else w.sp[ P.spill.oList + i] = r; }

// the document leaves the fast tier (a capacity of the spill area, or a case the compact state cannot express)
#define FALLBACK( WHY) do { w.err = SPD_FAST_FALLBACK; w.why = (WHY); } while (0)
enum {FB_LEXEMS=1, FB_DISPOSE=2, FB_ITEM_AFTER_RESULT=3, FB_ITEMS=4, FB_STAGED=5, FB_BUCKET_CHUNKS=6, FB_CHUNKS=7, FB_RULES=8, FB_BUCKET_SIZE=9};

// ---------------------------------------------------------------- rule word helpers
// end_ordpos <= pos (sequence / within guard) and == pos (sequence_imm), from the 8 low bits kept in the word;
// the true value lies in [pos-63, pos+1]
__device__ __forceinline__ bool endLE( u32 hw, u32 pos) { return (hw & H_ENDZERO) || ((hw >> H_END_SHIFT) & H_END_MASK) != ((pos + 1u) & 0xFFu); }
__device__ __forceinline__ bool endEQ( u32 hw, u32 pos) { return (hw & H_ENDZERO) ? (pos == 0) : (((hw >> H_END_SHIFT) & H_END_MASK) == (pos & 0xFFu)); }
__device__ __forceinline__ u32 withEnd( u32 hw, u32 end) { return (hw & ~((H_END_MASK << H_END_SHIFT) | H_ENDZERO)) | ((end & 0xFFu) << H_END_SHIFT); }

// ---------------------------------------------------------------- staged results
// {resultHandle, formatHandle, first lexem, last lexem, nItems | var0<<8 | var1<<16 | var2<<24, item lexems x3} (items latest first)
__device__ __forceinline__ void stageResult( Wave& w, KP P, u32 at, u32 handle, u32 fmt, u32 startLex, u32 endLex, u32 nItems, u32 vars, u32 i0, u32 i1, u32 i2)
{
	u32* S = &w.sp[ P.spill.oStaged + 8*at];
	st4( S, handle, fmt, startLex, endLex);
	st4( S+4, nItems | (vars << 8), i0, i1, i2);
}

// ---------------------------------------------------------------- fireSignal (cpp:772-979) on an installed trigger
// uniform: every lane computes the same; stores by lane 0
__device__ __forceinline__ void fireSignal( Wave& w, KP P, u32 tsv, u32 sord)
{
	const u32 tid = tsv & 0xFFFFu, r = tid >> 2;
	const u32 sigval = (tsv >> 16) & 0xFu, sigtype = (tsv >> 20) & 0x7u, hasVar = (tsv >> 23) & 1u, variable = tsv >> 24;
	u32 hw = bcast0( ldHot( w, P, r));
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
			if (LANE == 0) stHot( w, P, r, hw);
			if (w.nDispose < P.spill.maxRules) { if (LANE == 0) stList( w, P, w.nDispose, r); w.nDispose += 1; } else FALLBACK( FB_DISPOSE);
			return;
	}
	const bool done = (hw & H_DONE) != 0;
	u32 nItems = (hw >> H_NITEMS_SHIFT) & H_NITEMS_MASK;
	u32* cold = &w.sp[ P.spill.oCold + 8*r];		// {resultHandle, formatHandle, first taken lexem, item0, item1, item2, -, -}; item = lexem | variable<<24
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
			else if (nItems < 3u)
			{
				if (LANE == 0) cold[ 3+nItems] = w.lbase | (variable << 24);
				++nItems;
			}
			else { FALLBACK( FB_ITEMS); return; }
		}
		if (!(hw & H_HASSTART))
		{
			if (LANE == 0) cold[ 2] = w.lbase;
			if (sord) hw |= H_HASSTART;		// (a start at ordinal position 0 counts as unset, cpp:919)
		}
	}
	hw = (hw & ~(H_VALUE_MASK | (H_COUNT_MASK << H_COUNT_SHIFT) | (H_NITEMS_MASK << H_NITEMS_SHIFT))) | value | (count << H_COUNT_SHIFT) | (nItems << H_NITEMS_SHIFT);
	if (match && !done) hw |= H_DONE;
	if (LANE == 0) stHot( w, P, r, hw);
	if (match)
	{
		if (!done)
		{
			// the rule's write-once record: one round trip, nothing else is read from HBM on a match
			WAVE_FENCE();
			const uint4 c0 = ldu4( cold), c1 = ldu4( cold + 4);		// written by this wave (same-wave store -> load)
			if (c0.x)
			{
				if (w.nStaged < P.spill.maxStaged)
				{
					// items latest first
					const u32 ia = nItems == 3 ? c1.y : nItems == 2 ? c1.x : c0.w;
					const u32 ib = nItems == 3 ? c1.x : c0.w;
					const u32 ic = c0.w;
					const u32 vars = nItems == 0 ? 0u : nItems == 1 ? (ia >> 24) : nItems == 2 ? ((ia >> 24) | ((ib >> 24) << 8)) : ((ia >> 24) | ((ib >> 24) << 8) | ((ic >> 24) << 16));
					if (LANE == 0) stageResult( w, P, w.nStaged, c0.x, c0.y, c0.z, w.lbase, nItems, vars, ia & 0xFFFFFFu, ib & 0xFFFFFFu, ic & 0xFFFFFFu);
					w.nStaged += 1; w.nStagedItems += nItems;
				}
				else FALLBACK( FB_STAGED);
			}
		}
		if (fin)
		{
			if (w.nDispose < P.spill.maxRules) { if (LANE == 0) stList( w, P, w.nDispose, r); w.nDispose += 1; } else FALLBACK( FB_DISPOSE);
		}
	}
}

// ---------------------------------------------------------------- bucket chunks
// make room for the new sizes of the buckets (lane b < 16 holds newSize of bucket b)
__device__ __forceinline__ void reserveChunks( Wave& w, KP P, u32 newSize)
{
	u32 need = 0, have = 0;
	if (LANE < 16u) { have = w.bchunks[ LANE]; const u32 want = (newSize + FAST_CHUNK-1) / FAST_CHUNK; need = want > have ? want - have : 0u; }
	if (!__ballot( need != 0)) return;
	const bool tooMany = have + need > (u32)FAST_BUCKET_CHUNKS;
	u32 incl = waveScanAdd( need);
	const u32 total = (u32)__builtin_amdgcn_readlane( incl, 63);
	const u32 fromStack = w.cFreeN < total ? w.cFreeN : total;
	if (__ballot( tooMany)) { FALLBACK( FB_BUCKET_CHUNKS); return; }
	if (w.cUsed + (total - fromStack) > (u32)FAST_MAXCHUNKS) { FALLBACK( FB_CHUNKS); return; }
	const u32 excl = incl - need;
	for (u32 k=0; k<need; ++k)
	{
		const u32 q = excl + k;
		const u32 c = q < fromStack ? (u32)w.cfree[ w.cFreeN - 1 - q] : w.cUsed + (q - fromStack);
		w.ctab[ LANE*FAST_BUCKET_CHUNKS + have + k] = (u8)c;
	}
	if (need) w.bchunks[ LANE] = have + need;
	w.cFreeN -= fromStack; w.cUsed += total - fromStack;
	WAVE_FENCE();
}
// give back the chunks the buckets no longer need (called when the position advances)
__device__ __forceinline__ void trimChunks( Wave& w, KP P)
{
	u32 extra = 0, have = 0, want = 0;
	if (LANE < 16u) { have = w.bchunks[ LANE]; want = (w.bsize[ LANE] + FAST_CHUNK-1) / FAST_CHUNK; extra = have - want; }
	if (!__ballot( extra != 0)) return;
	u32 incl = waveScanAdd( extra);
	const u32 total = (u32)__builtin_amdgcn_readlane( incl, 63);
	const u32 excl = incl - extra;
	for (u32 k=0; k<extra; ++k) w.cfree[ w.cFreeN + excl + k] = w.ctab[ LANE*FAST_BUCKET_CHUNKS + want + k];
	if (extra) w.bchunks[ LANE] = want;
	w.cFreeN += total;
	WAVE_FENCE();
}

// ---------------------------------------------------------------- deactivation of a list of rules
// deactivateRule (cpp:679-702) for list[0..n) in list order; freeIds: disposeRule (cpp:704-708).
// The only order-dependent part is the swap-with-last removal of the rules' triggers from the 16 buckets:
// removals in different buckets do not interact, inside a bucket they must run in list order (rule by
// rule, a rule's triggers last installed first).  Every lane takes a rule; in each round a lane removes
// its next trigger if no lane before it still has a trigger in the same bucket.
__device__ __forceinline__ void deactivateList( Wave& w, KP P, u32 n, bool freeIds, bool mayRepeat)
{
	for (u32 base=0; base<n && !w.err; base+=64)
	{
		const u32 nb = (n - base) < 64u ? (n - base) : 64u;
		const bool have = LANE < nb;
		u32 r = 0, hw = 0;
		if (have) { r = ldList( w, P, base + LANE); hw = ldHot( w, P, r); }
		bool act = have && (hw & H_ACTIVE);
		if (mayRepeat && nb > 1)
		{
			// the same rule may be listed twice (deleted and finished in one transition): only its first entry acts
			for (u32 k=0; k+1<nb; ++k)
			{
				const u32 rk = (u32)__builtin_amdgcn_readlane( r, k);
				if (LANE > k && r == rk) act = false;
			}
		}
		u32 mask = act ? ((hw >> H_TMASK_SHIFT) & H_TMASK_MASK) : 0u;
		if (act) stHot( w, P, r, hw & ~(H_ACTIVE | (H_TMASK_MASK << H_TMASK_SHIFT)));
		// buckets of my triggers
		u32 hj[ 3];
#pragma unroll
		for (int j=0; j<3; ++j) hj[ j] = ((mask >> j) & 1u) ? (ldLink( w, P, r, (u32)j) >> 12) : 16u;
		u32 removed = 0;
#ifdef SPA_PROF
		const u64 prof_r0 = __builtin_amdgcn_s_memtime(); u32 prof_rounds = 0;
#endif
		for (u32 guard=0; guard<=3u*64u; ++guard)
		{
#ifdef SPA_PROF
			++prof_rounds;
#endif
			// my next trigger: the highest remaining slot
			const u32 j = mask ? (31u - (u32)__builtin_clz( mask)) : 0u;
			const u32 h = mask ? (j == 2 ? hj[ 2] : j == 1 ? hj[ 1] : hj[ 0]) : 0u;
			u32 pb = 0;
#pragma unroll
			for (int q=0; q<3; ++q) if ((mask >> q) & 1u) pb |= 1u << hj[ q];
			if (!__ballot( mask != 0)) break;
			const u32 before = fromLaneBelow( waveScanOr( pb));
			const bool go = mask != 0 && !((before >> h) & 1u);
			if (go)
			{
				// cpp:133-152: the bucket's last entry moves into the hole
				const u32 pos = ldLink( w, P, r, j) & 0xFFFu;
				const u32 last = w.bsize[ h] - 1u;
				if (pos != last)
				{
					const u32 li = entryIndex( w, h, last);
					const u32 me = ldEv( w, P, li), mt = ldTs( w, P, li);
					stEnt( w, P, entryIndex( w, h, pos), me, mt);
					stLink( w, P, (mt & 0xFFFFu) >> 2, mt & 3u, (h << 12) | pos);
				}
				w.bsize[ h] = last;
				mask &= ~(1u << j);
				++removed;
			}
			WAVE_FENCE();
		}
#ifdef SPA_PROF
		w.prof[ 7] += __builtin_amdgcn_s_memtime() - prof_r0;
		w.prof[ 8] += prof_rounds; w.prof[ 9] += 1; w.prof[ 10] += nb;
#endif
		{
			u32 incl = waveScanAdd( removed);
			w.nTrig -= (u32)__builtin_amdgcn_readlane( incl, 63);
		}
		if (freeIds)
		{
			const u64 mL = __ballot( have && r < w.R), mS = __ballot( have && r >= w.R);
			if (have)
			{
				if (r < w.R) w.freeS[ w.freeN + (u32)__popcll( mL & lanesBelow())] = (u16)r;
				else w.sp[ P.spill.oFree + w.sFreeN + (u32)__popcll( mS & lanesBelow())] = r - w.R;
			}
			w.freeN += (u32)__popcll( mL); w.sFreeN += (u32)__popcll( mS);
		}
		WAVE_FENCE();
	}
}

// ---------------------------------------------------------------- expiry (cpp:1084-1135, window part: ranges are <= 63)
__device__ __forceinline__ void setCurrentPos( Wave& w, KP P, u32 pos)
{
	if (w.curpos == pos) return;
	u32 wcnt = 0;
	for (; wcnt < 64u && w.curpos < pos && !w.err; ++wcnt, ++w.curpos)
	{
		const u32 slot = w.curpos & 63u;
		u32 r = ldu( &w.win[ slot]);
		if (r != (u32)NIL16)
		{
			// the rules of this position, last defined first (the list is LIFO like the reference's)
			u32 n = 0;
#ifdef SPA_PROF
			const u64 prof_w0 = __builtin_amdgcn_s_memtime();
#endif
			for (; r != (u32)NIL16; ++n)
			{
				if (n >= P.spill.maxRules) { w.err = SPD_ERR_INTERNAL; return; }
				if (LANE == 0) stList( w, P, n, r);
				r = bcast0( ldNext( w, P, r));
			}
#ifdef SPA_PROF
			w.prof[ 6] += __builtin_amdgcn_s_memtime() - prof_w0;
#endif
			if (LANE == 0) w.win[ slot] = (u16)NIL16;
			WAVE_FENCE();
			deactivateList( w, P, n, true, false);
		}
	}
	if (w.curpos < pos) w.curpos = pos;
	trimChunks( w, P);
}

// ---------------------------------------------------------------- installEventPrograms (cpp:1137-1157), 64 programs at a time
// Lane l instantiates program l of the key event's list.  Everything whose ORDER is observable lands where
// the sequential loop would have put it: bucket positions, expiry lists, results and dispose entries all
// advance in lane order through ballots and prefix sums.
struct Local		// a lexem event as a fired trigger sees it
{
	u32 sord, lex;
};

__device__ __forceinline__ void installBatch( Wave& w, KP P, u32 kb, u32 kc, u32 sord)
{
	for (u32 base=0; base<kc && !w.err; base+=64)
	{
		const u32 nb = (kc - base) < 64u ? (kc - base) : 64u;
		const bool have = LANE < nb;
		uint4 q0 = make_uint4( 0,0,0,0), q1 = q0, q2 = q0;
		if (have)
		{
			const FastKeyInst* K = &P.keyinst[ kb + base + LANE];
			q0 = ld4( K); q1 = ld4( (const u32*)K + 4); q2 = ld4( (const u32*)K + 8);
		}
		const u32 handle = q0.x, fmt = q0.y, pastEvent = q0.z, meta = q0.w;
		const u32 tEv[ 3] = {q1.x, q1.z, q2.x};
		u32 tInfo[ 3] = {q1.y, q1.w, q2.y};
		const u32 ntrig = (meta >> FKI_NTRIG_SHIFT) & FKI_NTRIG_MASK;
#pragma unroll
		for (int j=0; j<3; ++j) if ((u32)j >= ntrig) tInfo[ j] = 0;
		const u32 range = (meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK;
		// ---- signals on the fresh slot, in registers: (1) an alternative-keyed program replays the logged
		//      original key event (cpp:1253-1258 -> :1272-1334), (2) the key trigger(s) fire (cpp:1259-1269)
		u32 value = meta & FKI_VALUE_MASK, count = (meta >> FKI_COUNT_SHIFT) & FKI_COUNT_MASK, end = 0;
		u32 startLex = 0, nItems = 0, it0 = 0, it1 = 0, it2 = 0, nFires = 0;
		bool hasStart = false, done = false, fin = false, del = false, odd = false, resultNow = false;
		bool resultHasList = false;	// the result shares the rule's item list only if the list existed when the rule matched (cpp:941-953):
						// items captured later join that list (and show in the result), or start a list the result never sees
		auto fireLocal = [&]( u32 j, u32 info, const Local& e) {
			const u32 sigtype = (info >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK, sigval = info & FTI_SIGVAL_MASK;
			bool m = false, f = false, took = false;
			++nFires;
			switch (sigtype)
			{
				case SIG_ANY:
					took = true;
					if (count > 0) { m = true; --count; f = (count == 0); if (end < e.sord+1) end = e.sord+1; }
					break;
				case SIG_SEQUENCE:
				case SIG_SEQUENCE_IMM:
					if (sigval == value && (sigtype == SIG_SEQUENCE ? (end <= e.sord) : (end == e.sord)))
					{
						end = e.sord+1; value = sigval-1;
						if (count > 0) { --count; m = (count == 0); } else m = true;
						f = (value == 0); took = true;
					}
					break;
				case SIG_WITHIN:
					if ((sigval & value) != 0 && end <= e.sord)
					{
						end = e.sord+1; value &= ~sigval;
						if (count > 0) { --count; m = (count == 0); } else m = true;
						took = true;
					}
					break;
				default:	// SIG_DEL: the rule goes to the dispose list, nothing else happens (cpp:868-876)
					count = 0; value = 0; del = true;
					return;
			}
			if (took)
			{
				if ((info & FTI_HASVAR) && P.withItems)
				{
					if (done && !resultHasList) {}
					else if (nItems == 0) { it0 = e.lex | ((info >> FTI_VAR_SHIFT) << 24); nItems = 1; }
					else if (nItems == 1) { it1 = e.lex | ((info >> FTI_VAR_SHIFT) << 24); nItems = 2; }
					else if (nItems == 2) { it2 = e.lex | ((info >> FTI_VAR_SHIFT) << 24); nItems = 3; }
					else odd = true;
				}
				if (!hasStart) { startLex = e.lex; hasStart = (e.sord != 0); }
			}
			if (m) { if (!done) { done = true; resultNow = true; resultHasList = nItems != 0; } if (f) fin = true; }
		};
		bool dropped = false;		// deactivated by the replay: the rule never becomes visible to anything else
		const u32 psi = (meta >> FKI_PASTSTOP_SHIFT) & FKI_PASTSTOP_MASK;
		if (have && pastEvent && psi)
		{
			const u32 plex = w.stop[ 3*(psi-1)], psord = w.stop[ 3*(psi-1)+1], pts = w.stop[ 3*(psi-1)+2];
			if (pts && psord + range >= w.curpos)
			{
				Local pe; pe.sord = psord; pe.lex = plex;
#pragma unroll
				for (int j=2; j>=0; --j)		// the rule's trigger list: last installed first
				{
					if ((tInfo[ j] & FTI_INSTALL) && tEv[ j] == pastEvent) fireLocal( (u32)j, tInfo[ j], pe);
				}
				// a structure delimiter logged after the replayed event cancels the rule (cpp:1306-1321)
				bool cancelled = false;
#pragma unroll
				for (int j=0; j<3; ++j)
				{
					if ((tInfo[ j] & FTI_INSTALL) && ((tInfo[ j] >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) == SIG_DEL)
					{
						const u32 esi = (tInfo[ j] >> FTI_DELSTOP_SHIFT) & FTI_DELSTOP_MASK;
						if (esi) { const u32 ts = w.stop[ 3*(esi-1)+2]; if (ts && ts > pts) cancelled = true; }
					}
				}
				if (cancelled || del || fin) dropped = true;	// (replayPastEvent deactivates what its signals disposed, cpp:1322-1330)
			}
		}
		if (have && !dropped)
		{
			Local ke; ke.sord = sord; ke.lex = w.lbase;
#pragma unroll
			for (int j=0; j<3; ++j) if (tInfo[ j] & FTI_KEY) fireLocal( (u32)j, tInfo[ j], ke);
		}
		if (__ballot( odd)) { FALLBACK( FB_ITEMS); return; }
		const bool mat = have && !dropped;
		const u64 matMask = __ballot( mat);
		const u32 nmat = (u32)__popcll( matMask);
		const u32 rank = (u32)__popcll( matMask & lanesBelow());
		// ---- statistics (cpp:1251, :1256, :780)
		w.nInstalled += nb;
		w.nAlt += (u32)__popcll( __ballot( have && pastEvent != 0));
		{
			u32 incl = waveScanAdd( have ? nFires : 0u);
			w.nSignals += (u32)__builtin_amdgcn_readlane( incl, 63);
		}
		// ---- rule ids: LDS free stack, LDS bump, spill free stack, spill bump
		u32 r = 0;
		if (nmat)
		{
			const u32 a = w.freeN < nmat ? w.freeN : nmat;
			const u32 roomL = w.R - w.usedL;
			const u32 b = (nmat - a) < roomL ? (nmat - a) : roomL;
			const u32 c = w.sFreeN < (nmat - a - b) ? w.sFreeN : (nmat - a - b);
			const u32 d = nmat - a - b - c;
			if (w.R + w.usedS + d > P.spill.maxRules) { FALLBACK( FB_RULES); return; }
			if (mat)
			{
				if (rank < a) r = (u32)w.freeS[ w.freeN - 1 - rank];
				else if (rank < a+b) r = w.usedL + (rank - a);
				else if (rank < a+b+c) r = w.R + w.sp[ P.spill.oFree + w.sFreeN - 1 - (rank - a - b)];
				else r = w.R + w.usedS + (rank - a - b - c);
			}
			w.freeN -= a; w.usedL += b; w.sFreeN -= c; w.usedS += d;
		}
		// ---- expiry list of position sord+range (cpp:1066-1082): LIFO, the batch pushes in lane order
		if (nmat)
		{
			const u32 slot = (sord + range) & 63u;
			u64 same = matMask;
#pragma unroll
			for (int k=0; k<6; ++k)
			{
				const u64 bk = __ballot( mat && ((slot >> k) & 1u));
				same &= ((slot >> k) & 1u) ? bk : ~bk;
			}
			const u64 lower = same & lanesBelow();
			const u32 prevLane = lower ? (63u - (u32)__builtin_clzll( lower)) : 0u;
			const u32 prevRule = (u32)__builtin_amdgcn_ds_bpermute( (int)(prevLane << 2), (int)r);
			if (mat)
			{
				const u32 nx = lower ? prevRule : (u32)w.win[ slot];
				stNext( w, P, r, nx);
			}
			WAVE_FENCE();
			if (mat && !(same >> LANE >> 1)) w.win[ slot] = (u16)r;	// the last lane of a position becomes its head
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
			if (LANE < 16u) { oldSize = w.bsize[ LANE]; add = byteField( t0, t1, t2, t3, LANE); }
			if (__ballot( oldSize + add > 0xFFFu)) { FALLBACK( FB_BUCKET_SIZE); return; }
			reserveChunks( w, P, oldSize + add);
			if (w.err) return;
#pragma unroll
			for (int j=0; j<3; ++j)
			{
				if (inst[ j])
				{
					const u32 h = hB[ j];
					u32 sameB = 0;
#pragma unroll
					for (int jj=0; jj<j; ++jj) if (inst[ jj] && hB[ jj] == h) ++sameB;	// my own earlier templates
					const u32 pos = w.bsize[ h] + byteField( e0, e1, e2, e3, h) + sameB;
					const u32 sv = (tInfo[ j] & FTI_SIGVAL_MASK) | (((tInfo[ j] >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) << 4) | ((tInfo[ j] & FTI_HASVAR) ? 0x80u : 0u);
					stEnt( w, P, entryIndex( w, h, pos), tEv[ j], (4*r + (u32)j) | (sv << 16) | ((tInfo[ j] >> FTI_VAR_SHIFT) << 24));
					stLink( w, P, r, (u32)j, (h << 12) | pos);
					tmask |= 1u << j;
				}
			}
			WAVE_FENCE();
			if (LANE < 16u && add) w.bsize[ LANE] = oldSize + add;
			{
				const u32 s = t0 + t1 + t2 + t3;
				w.nTrig += (s & 0xFFu) + ((s >> 8) & 0xFFu) + ((s >> 16) & 0xFFu) + (s >> 24);	// (all fields together count <= 64 x 3 installs: no carries)
			}
		}
		// ---- the rule itself
		if (mat)
		{
			u32 hw = value | (count << H_COUNT_SHIFT) | H_ACTIVE | (tmask << H_TMASK_SHIFT) | (nItems << H_NITEMS_SHIFT);
			hw |= end ? ((end & 0xFFu) << H_END_SHIFT) : (u32)H_ENDZERO;
			if (done) hw |= H_DONE;
			if (hasStart) hw |= H_HASSTART;
			stHot( w, P, r, hw);
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
					const u32 o0 = ia & 0xFFFFFFu, o1 = ib & 0xFFFFFFu, o2 = ic & 0xFFFFFFu;
					stageResult( w, P, w.nStaged + (u32)__popcll( rm & lanesBelow()), handle, fmt, startLex, w.lbase, nItems, vars, o0, o1, o2);
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
				if (wantDispose) stList( w, P, w.nDispose + (u32)__popcll( dm & lanesBelow()), r);
				w.nDispose += nd;
			}
		}
		WAVE_FENCE();
	}
}

} // anonymous namespace

// ================================================================== kernel
extern "C" __global__ __launch_bounds__(64)
void spa_l2_fast_kernel( FastParams kernelArgs)
{
	KP P = kernelParams();
	extern __shared__ __attribute__((aligned(16))) u32 ldsRaw[];
	LDSQ u8* lb = (LDSQ u8*)ldsRaw;
	Wave w;
	w.sc = (lu32*)(lb + P.lds.oScalars); w.hot = (lu32*)(lb + P.lds.oHot);
	w.link = (lu16*)(lb + P.lds.oLink); w.next = (lu16*)(lb + P.lds.oNext); w.freeS = (lu16*)(lb + P.lds.oFree);
	w.ev = (lu32*)(lb + P.lds.oEv); w.ts = (lu32*)(lb + P.lds.oTs); w.ctab = (lu8*)(lb + P.lds.oChunkTab); w.cfree = (lu8*)(lb + P.lds.oChunkFree);
	w.bsize = (lu32*)(lb + P.lds.oBSize); w.bchunks = (lu32*)(lb + P.lds.oBChunks); w.win = (lu16*)(lb + P.lds.oWin);
	w.stop = (lu32*)(lb + P.lds.oStop); w.list = (lu16*)(lb + P.lds.oList);
	w.sp = P.spillBase + (u64)blockIdx.x * P.spill.totalWords;
	w.R = P.lds.R; w.T = P.lds.T;
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
		if (LANE < 16u) { w.bsize[ LANE] = 0; w.bchunks[ LANE] = 0; }
		w.win[ LANE] = (u16)NIL16;
		for (u32 s=LANE; s<P.nofStopWords; s+=64) w.stop[ 3*s+2] = 0;
		w.curpos = 0; w.timestamp = 0; w.nInstalled = 0; w.nAlt = 0; w.nSignals = 0; w.nTrig = 0; w.openLo = 0; w.openHi = 0;
		w.freeN = 0; w.usedL = 0; w.sFreeN = 0; w.usedS = 0; w.cFreeN = 0; w.cUsed = 0;
		w.nDispose = 0; w.nStaged = 0; w.nStagedItems = 0; w.err = 0; w.lbase = 0; w.why = 0;
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
			if (__ballot( kBegin == 0xFFFFFFFFu)) break;	// (the probes complete here)
			PROF_ADD( 5);
			for (u32 k=0; k<cnt && !w.err; ++k)
			{
				const u32 id = __builtin_amdgcn_readlane( lx.x, k), ordpos = __builtin_amdgcn_readlane( lx.y, k);
				const u32 origpos = __builtin_amdgcn_readlane( lx.z, k), origsize = __builtin_amdgcn_readlane( lx.w, k);
				const u32 origseg = __builtin_amdgcn_readlane( seg, k);
				// PatternMatcherContext::putInput (patternMatcher.cpp:131-162)
				if (curPosition > ordpos) { w.err = SPD_ERR_ORDER; break; }
				else if (curPosition < ordpos) { curPosition = ordpos; setCurrentPos( w, P, ordpos); PROF_ADD( 3); if (w.err) break; }
				else if (origsize >= 0x7FFFFFFFu || origseg >= 0x7FFFFFFFu || origpos >= 0x7FFFFFFFu) { w.err = SPD_ERR_RANGE; break; }
				if (id >= (1u<<29)) { w.err = SPD_ERR_RANGE; break; }
				w.lbase = (u32)(tile - lbeg) + k;
				// ---- doTransition (cpp:981-1064) for an input term: no follow events in a flat rule set
				{
					const u32 lo = w.openLo + w.nTrig;
					if (lo < w.openLo) w.openHi += 1;
					w.openLo = lo;
				}
				w.nDispose = 0;
				if (id)
				{
					// fire the triggers waiting for this event: 64 bucket entries per step, one ballot
					const u32 h = evhash( id) & 15u;
					const u32 n = ldu( &w.bsize[ h]);
					for (u32 base=0; base<n && !w.err; base+=64)
					{
						const u32 i = base + LANE;
						u32 idx = 0, e = 0;
						if (i < n) { idx = entryIndex( w, h, i); e = ldEv( w, P, idx); }
						u64 m = __ballot( i < n && e == id);
						while (m && !w.err)
						{
							const u32 p = (u32)__builtin_ctzll( m);
							m &= m-1;
							const u32 ip = (u32)__builtin_amdgcn_readlane( idx, p);
							const u32 tsv = bcast0( ldTs( w, P, ip));
							fireSignal( w, P, tsv, ordpos);
						}
					}
				}
				PROF_ADD( 0);
				if (w.err) break;
				// install the programs keyed by this event
				const u32 kb = __builtin_amdgcn_readlane( kBegin, k), kc = __builtin_amdgcn_readlane( kCount, k);
				const u32 stopIdx = __builtin_amdgcn_readlane( kStop, k);
				if (kc) installBatch( w, P, kb, kc, ordpos);
				PROF_ADD( 1);
				if (w.err) break;
				// deactivate rules that finished or were deleted
				if (w.nDispose) { WAVE_FENCE(); deactivateList( w, P, w.nDispose, false, true); }
				if (stopIdx)
				{
					if (LANE == 0) { w.stop[ 3*(stopIdx-1)] = w.lbase; w.stop[ 3*(stopIdx-1)+1] = ordpos; w.stop[ 3*(stopIdx-1)+2] = w.timestamp + 1u; }
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

namespace spa {
// resident single-wave workgroups per CU for a given LDS slice (registers and LDS both limit it)
int fastBlocksPerCU( unsigned ldsBytes)
{
	int n = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor( &n, spa_l2_fast_kernel, 64, ldsBytes) != hipSuccess || n < 1) n = 1;
	if (n > 32) n = 32;
	return n;
}
hipError_t launchL2Fast( const FastParams& P, unsigned nblocks, hipStream_t stream)
{
	hipLaunchKernelGGL( spa_l2_fast_kernel, dim3( nblocks), dim3( 64), P.lds.totalBytes, stream, P);
	return hipGetLastError();
}
}
