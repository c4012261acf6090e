// Level-2 rule automaton on gfx950: one wavefront per document.
//
// What it replaces (reference, CPU): StateMachine::doTransition / fireSignal / installProgram /
// setCurrentPos / replayPastEvent and PatternMatcherContext::putInput / fetchResults
// (src/ruleMatcherAutomaton.cpp:672-1334, src/patternMatcher.cpp:131-301).
//
// Execution model: a document is inherently sequential (every event mutates the rule state the
// next event sees), so the unit of parallelism is the document: each 64-lane wavefront (= one
// workgroup) owns one document at a time and fetches the next from a device-side cursor when done.
// Inside a document the control flow is wave-uniform (all lanes follow the same path on the same
// values, so branches are scalar and table/state reads are single-address broadcasts); the lanes are
// data-parallel workers where the algorithm has width: scanning a trigger bucket for an event id (the
// 64-lane analogue of the reference's SSE scan, src/ruleMatcherAutomaton.cpp:179-226), instantiating
// the programs of a key event, deactivating the rules that finished or expired, fetching lexems in
// coalesced 1 KiB rows, copying results out.
// Integer only: no MFMA.  The compiled ProgramTable (l2_tables.h) is read-only in HBM/L2; the mutable
// per-document state lives in a per-wave arena in HBM plus a small block of LDS (see below).
// The kernel is one function without calls: a called device function saves and restores the
// callee-saved vector registers it uses through scratch memory, which at one call per event was
// 50 KB of memory traffic per event (3.8x everything else together).
//
// Observable orders that are reproduced exactly (they decide which matches exist and in which
// order they are reported): order of triggers inside the 16 hash buckets incl. swap-with-last
// removal; LIFO order of the per-position dispose lists and of captured-item lists; libstdc++
// push_heap/pop_heap order of the far-expiry queue; order of follow events.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2_tables.h"
#include "l2_device.h"
#include "wave_scan.h"

using namespace spa;

namespace {

typedef uint32_t u32;
typedef uint64_t u64;

#define LANE ((u32)(threadIdx.x & 63u))
#ifndef SPA_L2_BATCH_MIN
#define SPA_L2_BATCH_MIN 1	/* key lists of at least this many programs are installed lane-parallel */
#endif
#ifndef SPA_L2_DEACT_MIN
#define SPA_L2_DEACT_MIN 2	/* dispose lists of at least this many rules are deactivated lane-parallel */
#endif
#ifndef SPA_L2_WAVES_PER_EU
#define SPA_L2_WAVES_PER_EU 3
#endif

// Debug build only (make TRACE=1): progress words written to host-mapped memory by wave 0 so a
// stuck kernel can be diagnosed from the host without waiting for it.
#ifdef SPA_TRACE
#define TRACE( SLOT, VALUE) do { if (P.trace && LANE == 0 && blockIdx.x == 0 && threadIdx.x < 64) { *(volatile u32*)&P.trace[ SLOT] = (u32)(VALUE); } } while (0)
#else
#define TRACE( SLOT, VALUE) do {} while (0)
#endif
// Phase profile (make PROF=1): wave-cycles per phase summed into counters[4..7]
#ifdef SPA_PROF
#define PROF_DECL u64 prof_t0 = __builtin_amdgcn_s_memtime()
#define PROF_ADD( SLOT) do { u64 prof_t1 = __builtin_amdgcn_s_memtime(); w.raw->prof[ SLOT] += prof_t1 - prof_t0; prof_t0 = prof_t1; } while (0)
#else
#define PROF_DECL do {} while (0)
#define PROF_ADD( SLOT) do {} while (0)
#endif
#ifdef SPA_PROF2
#define P2C( SLOT, N) do { w.raw->prof[ SLOT] += (N); } while (0)
#define P2_DECL u64 p2_t0 = __builtin_amdgcn_s_memtime()
#define P2_ADD( SLOT) do { u64 p2_t1 = __builtin_amdgcn_s_memtime(); w.raw->prof[ SLOT] += p2_t1 - p2_t0; p2_t0 = p2_t1; } while (0)
#else
#define P2C( SLOT, N) do {} while (0)
#define P2_DECL do {} while (0)
#define P2_ADD( SLOT) do {} while (0)
#endif
// capacity overflow of the per-document state (PROF2 builds remember where)
#ifdef SPA_PROF2
#define ARENA_FAIL do { w.err = SPD_ERR_ARENA; w.raw->prof[3] = __LINE__; } while (0)
#else
#define ARENA_FAIL w.err = SPD_ERR_ARENA
#endif
#ifdef SPA_TRACE2
#define TRACE2( SLOT, VALUE) TRACE( SLOT, VALUE)
#else
#define TRACE2( SLOT, VALUE) do {} while (0)
#endif

// The launch parameters are read where they are: in the kernel argument segment (constant address
// space, scalar loads through the scalar cache), also from the functions the kernel calls.
typedef const __attribute__((address_space(4))) L2Params& KP;
__device__ __forceinline__ KP kernelParams() { return *(const __attribute__((address_space(4))) L2Params*)__builtin_amdgcn_kernarg_segment_ptr(); }

enum {F_ACTIVE=1u, F_DONE=2u, F_EXT=4u};

struct EvData { u32 sseg, eseg, spos, epos, sord, eord, sub, fmt; };		// src/ruleMatcherAutomaton.hpp:200-218

// A rule instance and the triggers it has installed share one 128-byte block (one L2 cache line):
// installing, firing and deactivating a rule each touch that line and the bucket lines, nothing else.
// The per-document state of all resident waves is far larger than the L2, so the number of distinct
// lines an operation touches is what the kernel pays for.
struct Trig		// installed trigger (hpp:45-74, :118-127), 16 B
{
	u32 event, sigval, typevar, link;	// typevar = sigtype | variable<<4 ; link = bucket<<28 | position in the bucket
};
struct Rule		// rule instance + its action slot (hpp:87-104, :171-186) + 4 trigger slots, 128 B
{
	u32 value, count, flags, start_ordpos;
	u32 end_ordpos, start_origseg, start_origpos, program;
	u32 trigMask, dataRef, ext, owner;	// trigMask: occupied slots; ext: continuation block+1 (programs with more than 4 triggers); owner: rule of a continuation block
	u32 nInline, _pad[3];			// 1: one captured item sits in trigger slots 2 and 3 (see inlineItem)
	Trig trig[4];				// trigger id = 4*block + slot, filled in installation order
};
struct Item { u32 variable, next, _a, _b; EvData d; };	// captured variable (hpp:262-271), 48 B
struct Follow { EvData d; u32 event, _a, _b, _c; };	// 48 B
struct StopLog { EvData d; u32 timestamp, _a, _b, _c; };// 48 B
struct StagedResult { u32 program, sord, eord, sseg, spos, eseg, epos, dataRef; };	// 32 B; result and format handle are read from programs[program] at document end

// ---------------------------------------------------------------- where the per-document state lives
// Records and lists: a per-wave arena in HBM whose capacities grow on demand (layout: ArenaLayout).
// Scalars: a block of LDS per wave (struct WS).  A variant that kept rules, triggers and buckets in a
// 77 KB LDS slice (2 waves per CU) was measured slower than 8 arena waves per CU and was removed.
#define LDSQ __attribute__((address_space(3)))
#define HOT
#define CAP_RULES	(P.arena.maxRules)
#define CAP_BUCKET	(P.arena.bucketCap)
#define CAP_FOLLOW	(P.arena.maxFollow)
#define CAP_DISPOSE	(P.arena.maxDispose)
#define CAP_HEAP	(P.arena.maxHeap)
#define WIN_CHUNK	(P.arena.winChunk)
#define WIN_NCHUNKS	(P.arena.winChunks)
#define CAP_SCRATCH	(P.arena.scratchCap)
#define CAP_ITEMS	(P.arena.maxItems)
#define CAP_REFS	(P.arena.maxRefs)
typedef HOT u32 hu32;
typedef HOT Rule HRule;
typedef Trig HTrig;
typedef Item HItem;		// captured items and data references stay in HBM in both tiers: results keep them alive until the document ends
typedef HOT Follow HFollow;
typedef HOT EvData HEvData;

// Per-wave scalars: one block at the start of the workgroup's LDS (workgroups are single waves), so
// every function reaches them with a ds_read at a constant address instead of carrying ~30 live
// values through calls (which the register allocator would spill to scratch memory: 256 B of
// memory traffic per access).
struct WS
{
	u32 curpos, timestamp, nInstalled, nAlt, nSignals, nTrig;
	u64 open;
	u32 ruleFreeN, ruleUsed, itemFreeN, itemUsed, refFreeN, refUsed;
	u32 heapSize, nFollow, nDispose, nStaged, err;
	u32 winFreeN, winUsed;
#if defined(SPA_PROF) || defined(SPA_PROF2)
	u64 prof[4];
#endif
};
typedef LDSQ WS HWS;
typedef LDSQ u32 lu32;
typedef LDSQ u64 lu64;
// Access to one scalar of the wave state: reads come back through readfirstlane, so the value (and
// the branches and addresses computed from it) stays in scalar registers.
struct WSField
{
	lu32* p;
	__device__ __forceinline__ operator u32() const { return __builtin_amdgcn_readfirstlane( *p); }
	__device__ __forceinline__ u32 operator=( u32 v) const { *p = v; return v; }
	__device__ __forceinline__ u32 operator=( const WSField& o) const { u32 v = o; *p = v; return v; }
	__device__ __forceinline__ u32 operator+=( u32 v) const { u32 n = (u32)*this + v; *p = n; return n; }
	__device__ __forceinline__ u32 operator-=( u32 v) const { u32 n = (u32)*this - v; *p = n; return n; }
	__device__ __forceinline__ u32 operator++() const { return *this += 1u; }
	__device__ __forceinline__ u32 operator--() const { return *this -= 1u; }
	__device__ __forceinline__ u32 operator++( int) const { u32 o = *this; *p = o+1; return o; }
	__device__ __forceinline__ u32 operator--( int) const { u32 o = *this; *p = o-1; return o; }
};
struct WSField64
{
	lu64* p;
	__device__ __forceinline__ operator u64() const { u64 v = *p; return ((u64)__builtin_amdgcn_readfirstlane( (u32)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane( (u32)v); }
	__device__ __forceinline__ void operator=( u64 v) const { *p = v; }
	__device__ __forceinline__ void operator+=( u64 v) const { *p = (u64)*this + v; }
};
struct WSV		// view of the wave state block at `raw` (one LDS address; everything else folds to constants)
{
	HWS* raw;
	u32* arena;		// the wave's arena in HBM
	WSField curpos, timestamp, nInstalled, nAlt, nSignals, nTrig;
	WSField64 open;
	WSField ruleFreeN, ruleUsed, itemFreeN, itemUsed, refFreeN, refUsed;
	WSField heapSize, nFollow, nDispose, nStaged, err;
	WSField winFreeN, winUsed;
	__device__ __forceinline__ WSV( HWS* b, u32* a)
		:raw(b),arena(a),curpos{&b->curpos},timestamp{&b->timestamp},nInstalled{&b->nInstalled},nAlt{&b->nAlt},nSignals{&b->nSignals},nTrig{&b->nTrig}
		,open{&b->open},ruleFreeN{&b->ruleFreeN},ruleUsed{&b->ruleUsed}
		,itemFreeN{&b->itemFreeN},itemUsed{&b->itemUsed},refFreeN{&b->refFreeN},refUsed{&b->refUsed}
		,heapSize{&b->heapSize},nFollow{&b->nFollow},nDispose{&b->nDispose},nStaged{&b->nStaged},err{&b->err}
		,winFreeN{&b->winFreeN},winUsed{&b->winUsed} {}
};
typedef const WSV& WSR;

// the wave's arena in HBM (workgroup = one wave)
#define ARENA( OFS)	(w.arena + (OFS))
#define RULES		((Rule*)ARENA( P.arena.oRules))
#define TRIG( T)	(&RULES[ (T) >> 2].trig[ (T) & 3u])
#define BKT		ARENA( P.arena.oBEvent)		/* 16 buckets x bucketCap x {event, trigger id} */
// small, touched by every step: in LDS behind the wave state block
#define BSIZE		((lu32*)w.raw + 64)		/* 16 bucket sizes */
#define WINDOW		((lu32*)w.raw + 80)		/* 64 expiry list lengths */
#define WINCHUNK	((lu32*)w.raw + 144)		/* 64 x 8 chunk ids */
#define SCRLDS		((lu32*)w.raw + 656)		/* 16 x SCRLDS_CAP: bucket partition of a deactivation batch */
#define EXPLIST		((lu32*)w.raw + 1168)		/* EXPLIST_CAP: the rules expiring at the current position */
#define WINARR		ARENA( P.arena.oWinArr)
#define WINFREE		ARENA( P.arena.oWinFree)
#define SCRATCH		ARENA( P.arena.oScratch)
#define HEAP		ARENA( P.arena.oHeap)
#define FOLLOW		((Follow*)ARENA( P.arena.oFollow))
#define DISPOSE		ARENA( P.arena.oDispose)
#define RULEFREE	ARENA( P.arena.oRuleFree)
#define ITEMS		((Item*)ARENA( P.arena.oItems))
#define REFS		ARENA( P.arena.oRefs)
#define ITEMFREE	ARENA( P.arena.oItemFree)
#define REFFREE		ARENA( P.arena.oRefFree)
#define STOP		((StopLog*)ARENA( P.arena.oStop))
#define GSTACK		ARENA( P.arena.oGStack)
#define STAGED		((StagedResult*)ARENA( P.arena.oStaged))

__device__ __forceinline__ u32 evhash( u32 a)		// src/ruleMatcherAutomaton.cpp:34-40
{
	a += ~(a>>5);
	a +=  (a<<3);
	a ^=  (a>>4);
	return a;
}

__device__ __forceinline__ u32 bcast0( u32 v) { return __builtin_amdgcn_readfirstlane( v); }

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// Wave-uniform load: every lane reads the same address; the value is moved to a scalar register so
// that everything computed from it (indices, loop bounds, branch conditions) stays scalar and the
// control flow of the automaton is made of scalar branches, not exec-masked vector loops.
__device__ __forceinline__ u32 ldu( const u32* p) { return __builtin_amdgcn_readfirstlane( *p); }
// 16-byte accesses: one memory instruction moves a quarter/third/half of a record
__device__ __forceinline__ uint4 ld4( const void* p) { const u32x4 v = *(const u32x4*)p; return make_uint4( v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4( void* p, u32 a, u32 b, u32 c, u32 d) { u32x4 v; v.x = a; v.y = b; v.z = c; v.w = d; *(u32x4*)p = v; }
__device__ __forceinline__ u32* W( void* p) { return (u32*)p; }
__device__ __forceinline__ const u32* W( const void* p) { return (const u32*)p; }
__device__ __forceinline__ u32 ldu( const lu32* p) { return __builtin_amdgcn_readfirstlane( *p); }
__device__ __forceinline__ uint4 ld4( const LDSQ void* p) { const u32x4 v = *(const LDSQ u32x4*)p; return make_uint4( v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4( LDSQ void* p, u32 a, u32 b, u32 c, u32 d) { u32x4 v; v.x = a; v.y = b; v.z = c; v.w = d; *(LDSQ u32x4*)p = v; }
__device__ __forceinline__ lu32* W( LDSQ void* p) { return (lu32*)p; }
__device__ __forceinline__ const lu32* W( const LDSQ void* p) { return (const lu32*)p; }
template <class PTR>
__device__ __forceinline__ uint4 ldu4( PTR p)
{
	uint4 v = ld4( p);
	v.x = __builtin_amdgcn_readfirstlane( v.x); v.y = __builtin_amdgcn_readfirstlane( v.y);
	v.z = __builtin_amdgcn_readfirstlane( v.z); v.w = __builtin_amdgcn_readfirstlane( v.w);
	return v;
}
template <class PTR>
__device__ __forceinline__ void ldEv( EvData& d, PTR p)
{
	const uint4 a = ldu4( p), b = ldu4( W( p) + 4);
	d.sseg = a.x; d.eseg = a.y; d.spos = a.z; d.epos = a.w; d.sord = b.x; d.eord = b.y; d.sub = b.z; d.fmt = b.w;
}
template <class PTR>
__device__ __forceinline__ void stEv( PTR p, const EvData& d)
{
	st4( p, d.sseg, d.eseg, d.spos, d.epos);
	st4( W( p) + 4, d.sord, d.eord, d.sub, d.fmt);
}

// ---------------------------------------------------------------- allocators
// Which index a new record gets is not observable (only list orders are), so freed indices are
// kept on plain stacks: a whole batch of lanes can pop its indices with one gather.
__device__ __forceinline__ u32 allocRule( WSR w, KP P)
{
	u32 r;
	if (w.ruleFreeN) { r = ldu( &RULEFREE[ --w.ruleFreeN]); }
	else if (w.ruleUsed < CAP_RULES) { r = w.ruleUsed++; }
	else { ARENA_FAIL; r = 0; }
	return r;
}
__device__ __forceinline__ void freeRule( WSR w, KP P, u32 r)
{
	RULEFREE[ w.ruleFreeN++] = r;
}
__device__ __forceinline__ u32 allocItem( WSR w, KP P)
{
	u32 t;
	if (w.itemFreeN) { t = ldu( &ITEMFREE[ --w.itemFreeN]); }
	else if (w.itemUsed < CAP_ITEMS) { t = w.itemUsed++; }
	else { ARENA_FAIL; t = 0; }
	return t;
}
// data references: refs[2i] = head of the item list (1-based), refs[2i+1] = reference count
__device__ __forceinline__ u32 createRef( WSR w, KP P)	// cpp:750-753
{
	u32 t;
	if (w.refFreeN) { t = ldu( &REFFREE[ --w.refFreeN]); }
	else if (w.refUsed < CAP_REFS) { t = w.refUsed++; }
	else { ARENA_FAIL; return 0; }
	REFS[ 2*t] = 0; REFS[ 2*t+1] = 1;
	return t+1;
}
__device__ __forceinline__ void addRef( WSR w, KP P, u32 ref) { REFS[ 2*(ref-1)+1] = ldu( &REFS[ 2*(ref-1)+1]) + 1; }	// cpp:734-738

__device__ __forceinline__ void disposeRef( WSR w, KP P, u32 ref)				// cpp:710-732
{
	u32 cnt = ldu( &REFS[ 2*(ref-1)+1]);
	if (cnt > 1) { REFS[ 2*(ref-1)+1] = cnt-1; }
	else if (cnt == 1)
	{
		u32 it = ldu( &REFS[ 2*(ref-1)]);
		for (u32 guard=0; it; ++guard)
		{
			if (guard > w.itemUsed) { w.err = SPD_ERR_INTERNAL; break; }
			u32 nx = ldu( &ITEMS[ it-1].next);
			ITEMFREE[ w.itemFreeN++] = it-1;
			it = nx;
		}
		REFS[ 2*(ref-1)+1] = 0;
		REFFREE[ w.refFreeN++] = ref-1;
	}
	else { w.err = SPD_ERR_DATAREF; }
}

__device__ __forceinline__ void appendItem( WSR w, KP P, u32 ref, u32 variable, const EvData& d)	// cpp:740-748
{
	if (d.sub) addRef( w, P, d.sub);
	u32 it = allocItem( w, P);
	if (w.err) return;
	HItem* I = &ITEMS[ it];
	I->variable = variable; stEv( &I->d, d);
	I->next = ldu( &REFS[ 2*(ref-1)]);
	REFS[ 2*(ref-1)] = it+1;
}

__device__ __forceinline__ void joinItems( WSR w, KP P, u32 dest, u32 src)	// cpp:755-770
{
	u32 it = ldu( &REFS[ 2*(src-1)]);
	for (u32 guard=0; it && !w.err; ++guard)
	{
		if (guard > CAP_ITEMS) { w.err = SPD_ERR_INTERNAL; break; }
		HItem* S = &ITEMS[ it-1];
		u32 variable = ldu( &S->variable); EvData d; ldEv( d, &S->d); it = ldu( &S->next);
		appendItem( w, P, dest, variable, d);
	}
}

// ---------------------------------------------------------------- captured item kept in the rule block
// Until a rule completes (and a result or follow event has to share the list) the items it has
// captured are private to it, and most rule instances expire without completing.  The usual rule has
// at most two installed triggers and has captured one item by then: that item is kept in the two unused
// trigger slots of the rule's own cache line {variable,sseg,eseg,spos | epos,sord,eord,sub} (+ fmt in
// the padding) -- no item or reference record is allocated, linked, counted or released for it.  It
// moves to a shared list (materializeItems) when a second item arrives or the rule completes.
__device__ __forceinline__ void inlineItem( HRule* R, u32 variable, const EvData& d)
{
	st4( &R->trig[ 2], variable, d.sseg, d.eseg, d.spos);
	st4( &R->trig[ 3], d.epos, d.sord, d.eord, d.sub);
	R->_pad[ 0] = d.fmt;
	R->nInline = 1;
}
__device__ __forceinline__ u32 materializeItems( WSR w, KP P, HRule* R)
{
	const u32 ref = createRef( w, P);
	if (w.err) return 0;
	if (ldu( &R->nInline))
	{
		const u32 it = allocItem( w, P);
		if (w.err) return 0;
		HItem* I = &ITEMS[ it];
		const uint4 a = ldu4( &R->trig[ 2]), b = ldu4( &R->trig[ 3]);
		st4( I, a.x, 0, 0, 0);
		st4( W( I) + 4, a.y, a.z, a.w, b.x);
		st4( W( I) + 8, b.y, b.z, b.w, ldu( &R->_pad[ 0]));
		REFS[ 2*(ref-1)] = it+1;
		R->nInline = 0;
	}
	R->dataRef = ref;
	return ref;
}

// ---------------------------------------------------------------- event trigger table (cpp:114-257)
__device__ __forceinline__ void addTrigger( WSR w, KP P, u32 t, u32 event)
{
	u32 h = evhash( event) & 15u;
	u32 pos = ldu( &BSIZE[ h]);
	if (pos >= CAP_BUCKET) { ARENA_FAIL; return; }
	*(uint2*)&BKT[ 2*(h*CAP_BUCKET + pos)] = make_uint2( event, t);
	BSIZE[ h] = pos+1;
	TRIG( t)->link = (h << 28) | pos;
	w.nTrig += 1;
}
__device__ __forceinline__ void removeTrigger( WSR w, KP P, u32 t)	// swap with last, cpp:133-152
{
	u32 link = ldu( &TRIG( t)->link);
	u32 h = link >> 28, pos = link & 0x0FFFFFFFu;
	u32 last = ldu( &BSIZE[ h])-1;
	if (pos != last)
	{
		u32 me = ldu( &BKT[ 2*(h*CAP_BUCKET + last)]);
		u32 mi = ldu( &BKT[ 2*(h*CAP_BUCKET + last)+1]);
		*(uint2*)&BKT[ 2*(h*CAP_BUCKET + pos)] = make_uint2( me, mi);
		TRIG( mi)->link = link;
	}
	BSIZE[ h] = last;
	w.nTrig -= 1;
}

// The triggers of a rule, last installed first (the order of the reference's per-rule trigger list):
// continuation blocks from the deepest one back, then the slots 3..0 of the rule's own block.
template <class VISIT>
__device__ __forceinline__ void forEachTriggerLastFirst( WSR w, KP P, u32 r, VISIT visit)
{
	u32 depth = 0;
	for (u32 b=ldu( &RULES[ r].ext); b; b=ldu( &RULES[ b-1].ext))
	{
		if (++depth > CAP_RULES) { w.err = SPD_ERR_INTERNAL; return; }
	}
	for (u32 k=depth+1; k>0; --k)
	{
		u32 b = r;
		for (u32 i=1; i<k; ++i) b = ldu( &RULES[ b].ext)-1;
		const u32 mask = ldu( &RULES[ b].trigMask);
		for (int j=3; j>=0; --j) if ((mask >> j) & 1u) visit( 4*b + (u32)j);
	}
}

__device__ __forceinline__ void deactivateRule( WSR w, KP P, u32 r)	// cpp:679-702
{
	HRule* R = &RULES[ r];
	u32 flags = ldu( &R->flags);
	if (flags & F_ACTIVE)
	{
		R->flags = flags & ~F_ACTIVE;
		forEachTriggerLastFirst( w, P, r, [&]( u32 t) { removeTrigger( w, P, t); });
		for (u32 b=ldu( &R->ext); b; )			// release the continuation blocks
		{
			u32 nx = ldu( &RULES[ b-1].ext);
			RULES[ b-1].trigMask = 0; RULES[ b-1].ext = 0;
			RULEFREE[ w.ruleFreeN++] = b-1;
			b = nx;
		}
		R->trigMask = 0; R->ext = 0;
		u32 ref = ldu( &R->dataRef);
		if (ref) { disposeRef( w, P, ref); R->dataRef = 0; }
	}
}


__device__ __forceinline__ u64 lanesBelow() { return (1ull << LANE) - 1ull; }
__device__ __forceinline__ u32 byteSum( u32 v) { return (v & 0xFFu) + ((v >> 8) & 0xFFu) + ((v >> 16) & 0xFFu) + (v >> 24); }
__device__ __forceinline__ u32 byteField( u32 c0, u32 c1, u32 c2, u32 c3, u32 h)
{
	u32 wsel = h >> 2;
	u32 v = wsel == 0 ? c0 : wsel == 1 ? c1 : wsel == 2 ? c2 : c3;
	return (v >> ((h & 3u)*8)) & 0xFFu;
}

// ---------------------------------------------------------------- batched deactivation
// deactivateRule (cpp:679-702) for a list of rules, in list order.  The only order-dependent part
// is the swap-with-last removal of triggers from the 16 buckets; removals in different buckets do
// not interact, so after a stable partition of the triggers by bucket, lane b replays bucket b's
// removals in their original order while the other 15 buckets advance in the other lanes.
// Everything else (clearing the rule, releasing trigger / item / reference records) has no observable
// order and is done one rule per lane.
enum {DEACT_MAXCHAIN=4, SCRLDS_CAP=32, EXPLIST_CAP=128, REPLAY_MAX=8};

template <class LISTPTR>
__device__ __forceinline__ void deactivateBatch( HWS* wsBlock, u32* wsArena, KP P, LISTPTR list, u32 n, bool reversed, bool freeRules, bool checkDup)
{
	const WSV w( wsBlock, wsArena);
	for (u32 base=0; base<n && !w.err; base+=64)
	{
		P2_DECL;
		const u32 nb = (n - base) < 64u ? (n - base) : 64u;
		const bool have = LANE < nb;
		u32 r = 0;
		if (have) r = reversed ? list[ n - 1 - (base + LANE)] : list[ base + LANE];
		bool act = false;
		u32 mask = 0, ref = 0, ext = 0, flags = 0;
		if (have)
		{
			HRule* R = &RULES[ r];
			flags = R->flags;
			const uint4 q2 = ld4( W( R) + 8);	// {trigMask, dataRef, ext, owner}: same line, requested together with the flags
			act = (flags & F_ACTIVE) != 0;
			if (act) { mask = q2.x; ref = q2.y; ext = q2.z; }
		}
		// the data reference of my rule {list head, count}: requested now, used after the trigger work
		uint2 refRec = make_uint2( 0, 0);
		if (P.withItems && act && ref) refRec = *(const uint2*)&REFS[ 2*(ref-1)];
		if (checkDup)
		{
			// the same rule may be listed twice (deleted and finished in one step): only its first entry acts
			for (u32 k=0; k+1<nb; ++k)
			{
				u32 rk = (u32)__builtin_amdgcn_readlane( r, k);
				if (LANE > k && have && r == rk) act = false;
			}
			// ... and entries of earlier 64-blocks have already cleared F_ACTIVE in memory
		}
		if (act && !freeRules)		// (a block that is released below is dead: nothing reads it before its next installation rewrites it)
		{
			HRule* R = &RULES[ r];
			R->flags = flags & ~F_ACTIVE;
			*(uint2*)&R->trigMask = make_uint2( 0, 0);	// trigMask, dataRef
		}
		// my triggers, last installed first: the occupied slots of my block from the top
		u32 t[ DEACT_MAXCHAIN]; u32 nt = 0;
		const bool longChain = act && ext != 0;		// continuation blocks (more than 4 triggers): rare
		{
			u32 rem = act ? mask : 0u;
#pragma unroll
			for (int c=0; c<DEACT_MAXCHAIN; ++c)
			{
				t[ c] = 0;
				if (rem) { const u32 j = 31u - (u32)__builtin_clz( rem); rem &= ~(1u << j); t[ c] = 4*r + j; ++nt; }
			}
		}
		if (__ballot( longChain))
		{
			// rare: finish this block one rule at a time (flags were cleared above: restore, then serial)
			if (act && !freeRules) { HRule* R = &RULES[ r]; R->flags = flags; R->trigMask = mask; R->dataRef = ref; }
			__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
			for (u32 k=0; k<nb && !w.err; ++k)
			{
				u32 rk = (u32)__builtin_amdgcn_readlane( r, k);
				deactivateRule( w, P, rk);
				if (freeRules) freeRule( w, P, rk);
			}
			continue;
		}
		// stable partition of the triggers by bucket: scratch[b*cap + rank]
		u32 hOf[ DEACT_MAXCHAIN];
#pragma unroll
		for (int c=0; c<DEACT_MAXCHAIN; ++c) hOf[ c] = ((u32)c < nt) ? (TRIG( t[ c])->link >> 28) : 16u;
		if (__ballot( hOf[ 0] == 77u)) return;	// (forces the loads above to complete here when timing)
		P2_ADD( 0);
		u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
		for (int c=0; c<DEACT_MAXCHAIN; ++c)
		{
			if (hOf[ c] < 16u)
			{
				u32 inc = 1u << ((hOf[ c] & 3u)*8), ws = hOf[ c] >> 2;
				if (ws == 0) c0 += inc; else if (ws == 1) c1 += inc; else if (ws == 2) c2 += inc; else c3 += inc;
			}
		}
		u32 i0 = c0, i1 = c1, i2 = c2, i3 = c3;
		// 64 lanes x DEACT_MAXCHAIN = 256 could overflow an 8-bit field: scan in 16-bit halves
		u32 lo0 = i0 & 0x00FF00FFu, hi0 = (i0 >> 8) & 0x00FF00FFu, lo1 = i1 & 0x00FF00FFu, hi1 = (i1 >> 8) & 0x00FF00FFu;
		u32 lo2 = i2 & 0x00FF00FFu, hi2 = (i2 >> 8) & 0x00FF00FFu, lo3 = i3 & 0x00FF00FFu, hi3 = (i3 >> 8) & 0x00FF00FFu;
		lo0 = waveScanAdd( lo0); hi0 = waveScanAdd( hi0); lo1 = waveScanAdd( lo1); hi1 = waveScanAdd( hi1);
		lo2 = waveScanAdd( lo2); hi2 = waveScanAdd( hi2); lo3 = waveScanAdd( lo3); hi3 = waveScanAdd( hi3);
		// field h of word ws: h&3 == 0 -> lo bits 0..15, 1 -> hi bits 0..15, 2 -> lo bits 16..31, 3 -> hi bits 16..31
		auto incl = [&]( u32 h) -> u32 {
			u32 ws = h >> 2, f = h & 3u;
			u32 lo = ws == 0 ? lo0 : ws == 1 ? lo1 : ws == 2 ? lo2 : lo3;
			u32 hi = ws == 0 ? hi0 : ws == 1 ? hi1 : ws == 2 ? hi2 : hi3;
			u32 v = (f & 1u) ? hi : lo;
			return (f & 2u) ? (v >> 16) : (v & 0xFFFFu);
		};
		auto own = [&]( u32 h) -> u32 { return byteField( c0, c1, c2, c3, h); };
		// totals per bucket = inclusive values of lane 63; lane b < 16 owns bucket b
		u32 myCount = 0;
		{
			const u32 tl0 = (u32)__builtin_amdgcn_readlane( lo0, 63), th0 = (u32)__builtin_amdgcn_readlane( hi0, 63);
			const u32 tl1 = (u32)__builtin_amdgcn_readlane( lo1, 63), th1 = (u32)__builtin_amdgcn_readlane( hi1, 63);
			const u32 tl2 = (u32)__builtin_amdgcn_readlane( lo2, 63), th2 = (u32)__builtin_amdgcn_readlane( hi2, 63);
			const u32 tl3 = (u32)__builtin_amdgcn_readlane( lo3, 63), th3 = (u32)__builtin_amdgcn_readlane( hi3, 63);
			if (LANE < 16u)
			{
				const u32 ws = LANE >> 2, f = LANE & 3u;
				const u32 lo = ws == 0 ? tl0 : ws == 1 ? tl1 : ws == 2 ? tl2 : tl3;
				const u32 hi = ws == 0 ? th0 : ws == 1 ? th1 : ws == 2 ? th2 : th3;
				const u32 v = (f & 1u) ? hi : lo;
				myCount = (f & 2u) ? (v >> 16) : (v & 0xFFFFu);
			}
		}
		// the partition lives in LDS when every bucket's share fits (the usual case), else in the arena
		const bool scrInLds = !__ballot( myCount > (u32)SCRLDS_CAP);
		const u32 scap = CAP_SCRATCH;
		u32 seen0 = 0, seen1 = 0, seen2 = 0, seen3 = 0;	// my own earlier triggers per bucket
		u32 ntot = 0;
#pragma unroll
		for (int c=0; c<DEACT_MAXCHAIN; ++c)
		{
			if (hOf[ c] < 16u)
			{
				const u32 h = hOf[ c];
				const u32 mineBefore = byteField( seen0, seen1, seen2, seen3, h);
				const u32 rankInBucket = incl( h) - own( h) + mineBefore;
				if (scrInLds) SCRLDS[ h*SCRLDS_CAP + rankInBucket] = t[ c];
				else if (rankInBucket < scap) SCRATCH[ h*scap + rankInBucket] = t[ c]; else ARENA_FAIL;
				u32 inc = 1u << ((h & 3u)*8), ws = h >> 2;
				if (ws == 0) seen0 += inc; else if (ws == 1) seen1 += inc; else if (ws == 2) seen2 += inc; else seen3 += inc;
				++ntot;
			}
		}
		if (__ballot( w.err != 0)) { ARENA_FAIL; return; }
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		P2_ADD( 1);
		// lane b replays the removals of bucket b (cpp:133-152): each removal moves the bucket's current
		// last entry into the hole.  Only the last m entries can ever move, so for m <= REPLAY_MAX the lane
		// fetches them and the m links once (all loads in flight together) and plays the sequence in
		// registers: no memory round trip per removal.  Entries that end up below the new size are
		// stored (with the link of the moved trigger); everything at or above it is dead.
		{
			if (myCount)
			{
				const u32 b = LANE;
				uint2* bk = (uint2*)BKT + b*CAP_BUCKET;		// {event, trigger id}
				u32 n = BSIZE[ b];
				for (u32 done=0; done<myCount; done+=REPLAY_MAX)	// REPLAY_MAX removals per round of loads
				{
					const u32 m = (myCount - done) < (u32)REPLAY_MAX ? (myCount - done) : (u32)REPLAY_MAX;
					u32 tk[ REPLAY_MAX], lk[ REPLAY_MAX], ce[ REPLAY_MAX], ct[ REPLAY_MAX], mt[ REPLAY_MAX], mp[ REPLAY_MAX];
#pragma unroll
					for (int k=0; k<REPLAY_MAX; ++k)
					{
						tk[ k] = 0xFFFFFFFFu;
						if ((u32)k < m) tk[ k] = scrInLds ? SCRLDS[ b*SCRLDS_CAP + done + k] : SCRATCH[ b*scap + done + k];
					}
#pragma unroll
					for (int k=0; k<REPLAY_MAX; ++k)
					{
						lk[ k] = 0; ce[ k] = 0; ct[ k] = 0xFFFFFFFEu; mt[ k] = 0xFFFFFFFDu; mp[ k] = 0;
						if ((u32)k < m)
						{
							lk[ k] = TRIG( tk[ k])->link & 0x0FFFFFFFu;
							const uint2 e = bk[ n-1-(u32)k];	// the entry at position n-1-k
							ce[ k] = e.x; ct[ k] = e.y;
						}
					}
#pragma unroll
					for (int k=0; k<REPLAY_MAX; ++k)
					{
						if ((u32)k < m)
						{
							// where is trigger tk[k] now?  its stored link, unless this round has moved it
							u32 pos = lk[ k];
#pragma unroll
							for (int i=0; i<k; ++i) if (mt[ i] == tk[ k]) pos = mp[ i];			// moved below the tail
#pragma unroll
							for (int j=k; j<REPLAY_MAX; ++j) if (ct[ j] == tk[ k]) pos = n-1-(u32)j;	// sits in the tail
							const u32 last = n-1-(u32)k;
							if (pos != last)
							{
								const u32 me = ce[ k], mi = ct[ k];		// the entry at `last` moves into the hole
								if (pos >= n-m)
								{
									const u32 jj = n-1-pos;			// a hole inside the tail: stays in registers
#pragma unroll
									for (int j=k+1; j<REPLAY_MAX; ++j) if ((u32)j == jj) { ce[ j] = me; ct[ j] = mi; }
								}
								else
								{
									bk[ pos] = make_uint2( me, mi);
									TRIG( mi)->link = (b << 28) | pos;
									mt[ k] = mi; mp[ k] = pos;
								}
							}
						}
					}
					n -= m;
				}
				BSIZE[ b] = n;
			}
		}
		if (__ballot( BSIZE[ LANE & 15u] == 0x7FFFFFFFu)) return;
		P2_ADD( 3);
		// the trigger slots go with their rule blocks: only the count of installed triggers changes
		w.nTrig -= (u32)__popcll( __ballot( ntot & 1u)) + 2u*(u32)__popcll( __ballot( ntot & 2u)) + 4u*(u32)__popcll( __ballot( ntot & 4u));
		// release the data references (cpp:696-700 -> :710-732): one rule per lane
		if (P.withItems)
		{
			bool bad = false;
			u32 nfree = 0; bool freeRef = false;
			u32 itq[ 4] = {0,0,0,0};		// the first items of my list (usually all of them)
			if (act && ref)
			{
				const u32 cnt = refRec.y;
				if (cnt > 1) REFS[ 2*(ref-1)+1] = cnt-1;
				else if (cnt == 1)
				{
					freeRef = true;
					u32 it = refRec.x;
#pragma unroll
					for (int q=0; q<4; ++q) { if (it) { itq[ q] = it; it = ITEMS[ it-1].next; ++nfree; } }
					for (u32 g=0; it; it=ITEMS[ it-1].next, ++g) { ++nfree; if (g > w.itemUsed) { bad = true; break; } }
					REFS[ 2*(ref-1)+1] = 0;
				}
				else bad = true;
			}
			if (__ballot( bad)) { w.err = SPD_ERR_DATAREF; return; }
			u32 incI = nfree;
			incI = waveScanAdd( incI);
			const u32 totalI = (u32)__builtin_amdgcn_readlane( incI, 63);
			if (freeRef)
			{
				u32 at = w.itemFreeN + incI - nfree;
#pragma unroll
				for (int q=0; q<4; ++q) if (itq[ q]) ITEMFREE[ at++] = itq[ q]-1;
				if (nfree > 4) for (u32 it=ITEMS[ itq[ 3]-1].next; it; ) { u32 nx = ITEMS[ it-1].next; ITEMFREE[ at++] = it-1; it = nx; }
			}
			const u64 fm = __ballot( freeRef);
			if (freeRef) REFFREE[ w.refFreeN + (u32)__popcll( fm & lanesBelow())] = ref-1;
			w.itemFreeN += totalI;
			w.refFreeN += (u32)__popcll( fm);
		}
		P2_ADD( 2);
		if (freeRules)
		{
			if (have) RULEFREE[ w.ruleFreeN + LANE] = r;
			w.ruleFreeN += nb;
		}
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
	}
}

// ---------------------------------------------------------------- expiry (cpp:1066-1135)
// far-expiry queue: binary heap on `pos` (min-heap via the inverted comparison of hpp:425-428),
// sifted exactly like libstdc++'s __push_heap/__adjust_heap so ties come out in the same order.
__device__ __forceinline__ void heapPush( WSR w, KP P, u32 pos, u32 idx)
{
	if (w.heapSize >= CAP_HEAP) { ARENA_FAIL; return; }
	u32 hole = w.heapSize++;
	while (hole > 0)
	{
		u32 parent = (hole-1) >> 1;
		u32 ppos = ldu( &HEAP[ 2*parent]);
		if (!(ppos > pos)) break;			// comp(parent, value) == parent.pos > value.pos
		HEAP[ 2*hole] = ppos; HEAP[ 2*hole+1] = ldu( &HEAP[ 2*parent+1]);
		hole = parent;
	}
	HEAP[ 2*hole] = pos; HEAP[ 2*hole+1] = idx;
}
__device__ __forceinline__ void heapPop( WSR w, KP P)
{
	u32 n = w.heapSize;
	if (n > 1)
	{
		u32 len = n-1;
		u32 vpos = ldu( &HEAP[ 2*len]), vidx = ldu( &HEAP[ 2*len+1]);
		u32 hole = 0, child = 0;
		while (child < (len-1)/2)
		{
			child = 2*(child+1);
			if (ldu( &HEAP[ 2*child]) > ldu( &HEAP[ 2*(child-1)])) child--;	// comp(first[child], first[child-1])
			HEAP[ 2*hole] = ldu( &HEAP[ 2*child]); HEAP[ 2*hole+1] = ldu( &HEAP[ 2*child+1]);
			hole = child;
		}
		if ((len & 1) == 0 && child == (len-2)/2)
		{
			child = 2*(child+1);
			HEAP[ 2*hole] = ldu( &HEAP[ 2*(child-1)]); HEAP[ 2*hole+1] = ldu( &HEAP[ 2*(child-1)+1]);
			hole = child-1;
		}
		while (hole > 0)
		{
			u32 parent = (hole-1) >> 1;
			u32 ppos = ldu( &HEAP[ 2*parent]);
			if (!(ppos > vpos)) break;
			HEAP[ 2*hole] = ppos; HEAP[ 2*hole+1] = ldu( &HEAP[ 2*parent+1]);
			hole = parent;
		}
		HEAP[ 2*hole] = vpos; HEAP[ 2*hole+1] = vidx;
	}
	w.heapSize = n-1;
}

// ---- expiry window (cpp:1066-1082): the rules that expire at one of the next 64 positions, per
// position in definition order.  A position's list is a sequence of fixed-size chunks taken from a
// pool (a rule sits in exactly one list, so the pool is bounded by the number of live rules, while
// a single position may receive a whole batch of 64 installs at once).
__device__ __forceinline__ u32 winAllocChunk( WSR w, KP P)
{
	if (w.winFreeN) return ldu( &WINFREE[ --w.winFreeN]);
	if (w.winUsed < WIN_NCHUNKS) return w.winUsed++;
	ARENA_FAIL;
	return 0;
}
// make room for `n` more entries in the list of `slot` (currently `cnt` entries)
__device__ __forceinline__ void winReserve( WSR w, KP P, u32 slot, u32 cnt, u32 n)
{
	const u32 C = WIN_CHUNK;
	u32 have = (cnt + C-1) / C;
	const u32 need = (cnt + n + C-1) / C;
	if (need > 8u) { ARENA_FAIL; return; }
	for (; have < need && !w.err; ++have) { const u32 c = winAllocChunk( w, P); WINCHUNK[ slot*8 + have] = c; }
}
__device__ __forceinline__ u32 winEntryIndex( WSR w, KP P, u32 slot, u32 i)	// may be called per lane
{
	const u32 C = WIN_CHUNK;
	return WINCHUNK[ slot*8 + i/C]*C + (i % C);
}
__device__ __forceinline__ void winAppendOne( WSR w, KP P, u32 slot, u32 r)
{
	const u32 cnt = ldu( &WINDOW[ slot]);
	winReserve( w, P, slot, cnt, 1);
	if (w.err) return;
	WINARR[ winEntryIndex( w, P, slot, cnt)] = r;
	WINDOW[ slot] = cnt+1;
}
// copy the list of `slot` to EXPLIST (LDS) or, when it is longer, to DISPOSE[0..cnt) (free while no
// transition is running) and release its chunks
__device__ __forceinline__ void winTake( WSR w, KP P, u32 slot, u32 cnt)
{
	if (cnt <= (u32)EXPLIST_CAP)
	{
		for (u32 i=LANE; i<cnt; i+=64) EXPLIST[ i] = WINARR[ winEntryIndex( w, P, slot, i)];
	}
	else
	{
		if (cnt > CAP_DISPOSE) { ARENA_FAIL; return; }
		for (u32 i=LANE; i<cnt; i+=64) DISPOSE[ i] = WINARR[ winEntryIndex( w, P, slot, i)];
	}
	const u32 C = WIN_CHUNK, nc = (cnt + C-1) / C;
	if (LANE < nc) WINFREE[ w.winFreeN + LANE] = WINCHUNK[ slot*8 + LANE];
	w.winFreeN += nc;
	WINDOW[ slot] = 0;
}

__device__ __forceinline__ void defineDisposeRule( WSR w, KP P, u32 pos, u32 r)	// cpp:1066-1082
{
	// pos >= curpos always holds here: installProgram has rejected expired programs
	if (pos < w.curpos + 64u) winAppendOne( w, P, pos & 63u, r);
	else heapPush( w, P, pos, r);
}

__device__ __forceinline__ void disposeRule( WSR w, KP P, u32 r)	// cpp:704-708
{
	deactivateRule( w, P, r);
	freeRule( w, P, r);
}

__device__ __forceinline__ void setCurrentPos( WSR w, KP P, u32 pos)	// cpp:1084-1135
{
	if (w.curpos == pos) return;
	u32 wcnt = 0;
	for (; wcnt < 64u && w.curpos < pos; ++wcnt, ++w.curpos)
	{
		u32 widx = w.curpos & 63u;
		if (widx == 0)
		{
			while (w.heapSize && ldu( &HEAP[0]) < w.curpos + 64u)
			{
				wcnt = 0;
				u32 hp = ldu( &HEAP[0]), hr = ldu( &HEAP[1]);
				winAppendOne( w, P, hp & 63u, hr);
				if (w.err) return;
				heapPop( w, P);
			}
		}
		const u32 cnt = ldu( &WINDOW[ widx]);
		if (cnt)
		{
			// the rules of this position, last defined first (the reference's LIFO list order)
			winTake( w, P, widx, cnt);
			if (w.err) return;
			if (cnt <= (u32)EXPLIST_CAP) deactivateBatch( w.raw, w.arena, P, EXPLIST, cnt, true/*reversed*/, true/*free the rules*/, false);
			else deactivateBatch( w.raw, w.arena, P, DISPOSE, cnt, true/*reversed*/, true/*free the rules*/, false);
		}
	}
	if (w.curpos < pos)
	{
		w.curpos = pos;
		while (w.heapSize && ldu( &HEAP[0]) < w.curpos)
		{
			u32 hr = ldu( &HEAP[1]);
			heapPop( w, P);
			disposeRule( w, P, hr);
		}
	}
}

// ---------------------------------------------------------------- fireSignal (cpp:772-979)
__device__ __forceinline__ void fireSignal( WSR w, KP P, u32 r, u32 sigtype, u32 sigval, u32 variable, const EvData& d)
{
	HRule* R = &RULES[ r];
	uint4 q0 = ldu4( R), q1 = ldu4( W( R) + 4);		// {value,count,flags,start_ordpos} {end_ordpos,start_origseg,start_origpos,program}
	if (q0.z & F_EXT)
	{
		// trigger slot in a continuation block: the state is in the owner's block
		r = ldu( &R->owner); R = &RULES[ r];
		q0 = ldu4( R); q1 = ldu4( W( R) + 4);
	}
	u32 value = q0.x, count = q0.y, flags = q0.z, end_ordpos = q1.x;
	bool match = false, take = false, fin = false;
	w.nSignals += 1;

	switch (sigtype)
	{
		case SIG_ANY:
			take = true;
			if (count > 0)
			{
				match = true; --count; fin = (count == 0);
				if (end_ordpos < d.eord) end_ordpos = d.eord;
			}
			break;
		case SIG_AND:
			if (count > 0)
			{
				if (!value)
				{
					value = d.sord;
					if (end_ordpos > d.eord) end_ordpos = d.eord;
				}
				if (value == d.sord) { match = true; --count; fin = (count == 0); take = true; }
			}
			break;
		case SIG_SEQUENCE:
		case SIG_SEQUENCE_IMM:
		{
			bool posok = (sigtype == SIG_SEQUENCE) ? (end_ordpos <= d.sord) : (end_ordpos == d.sord);
			if (sigval == value && posok)
			{
				end_ordpos = d.eord; value = sigval-1;
				if (count > 0) { --count; match = (count == 0); } else match = true;
				fin = (value == 0); take = true;
			}
			break;
		}
		case SIG_WITHIN:
			if ((sigval & value) != 0 && end_ordpos <= d.sord)
			{
				end_ordpos = d.eord; value &= ~sigval;
				if (count > 0) { --count; match = (count == 0); } else match = true;
				fin = (value == 0); take = true;
			}
			break;
		default: // SIG_DEL
			R->count = 0; R->value = 0;
			if (w.nDispose < CAP_DISPOSE) DISPOSE[ w.nDispose++] = r; else ARENA_FAIL;
			return;
	}
	u32 start_ordpos = q0.w, start_origseg = q1.y, start_origpos = q1.z;
	u32 dataRef = ldu( &R->dataRef);
	if (take)
	{
		if (P.withItems)
		{
			if (variable)
			{
				if (!dataRef)
				{
					const uint4 q2 = ldu4( W( R) + 8), q3 = ldu4( W( R) + 12);	// {trigMask,dataRef,ext,owner} {nInline,..}
					if (!q3.x && !(q2.x & 0xCu) && !q2.z)
					{
						if (d.sub) addRef( w, P, d.sub);
						inlineItem( R, variable, d);
					}
					else
					{
						dataRef = materializeItems( w, P, R);
						if (!w.err) appendItem( w, P, dataRef, variable, d);
					}
				}
				else appendItem( w, P, dataRef, variable, d);
			}
			else if (d.sub)
			{
				if (!dataRef) dataRef = materializeItems( w, P, R);
				if (!w.err) joinItems( w, P, dataRef, d.sub);
			}
		}
		if (start_ordpos == 0)
		{
			start_ordpos = d.sord; start_origseg = d.sseg; start_origpos = d.spos;
		}
		else if (start_ordpos > d.sord)
		{
			start_ordpos = d.sord;
			if (start_origseg > d.sseg || (start_origseg == d.sseg && start_origpos > d.spos))
			{
				start_origseg = d.sseg; start_origpos = d.spos;
			}
		}
	}
	const u32 newFlags = (match && !(flags & F_DONE)) ? (flags | F_DONE) : flags;
	st4( R, value, count, newFlags, start_ordpos);
	st4( W( R) + 4, end_ordpos, start_origseg, start_origpos, q1.w);
	if (match)
	{
		if (!(flags & F_DONE))
		{
			const uint4 g0 = ldu4( &P.programs[ q1.w]), g1 = ldu4( W( &P.programs[ q1.w]) + 4);	// {initsigval,initcount,event,resultHandle} {formatHandle,..}
			u32 fevent = g0.z, handle = g0.w, fmt = g1.x;
			// a follow event / result shares the captured items from here on: an item kept in the block moves to a list
			if (P.withItems && !dataRef && (fevent || handle) && ldu( &R->nInline)) dataRef = materializeItems( w, P, R);
			if (fevent)
			{
				if (w.nFollow < CAP_FOLLOW)
				{
					HFollow* F = &FOLLOW[ w.nFollow++];
					st4( F, start_origseg, d.eseg, start_origpos, d.epos);
					st4( W( F) + 4, start_ordpos, end_ordpos, dataRef, fmt);
					F->event = fevent;
					if (dataRef) addRef( w, P, dataRef);
				}
				else ARENA_FAIL;
			}
			if (handle)
			{
				if (w.nStaged < P.arena.maxStaged)
				{
					StagedResult* S = &STAGED[ w.nStaged++];
					st4( S, q1.w, start_ordpos, end_ordpos, start_origseg);
					st4( W( S) + 4, start_origpos, d.eseg, d.epos, dataRef);
					if (dataRef) addRef( w, P, dataRef);
				}
				else ARENA_FAIL;
			}
		}
		if (fin)
		{
			if (w.nDispose < CAP_DISPOSE) DISPOSE[ w.nDispose++] = r; else ARENA_FAIL;
		}
	}
}

__device__ __forceinline__ const DevKeyEntry* lookupKey( KP P, u32 event)
{
	if (!event) return 0;
	u32 slot = keyHash( event) & P.keymask;
	for (u32 probes=0; probes<=P.keymask; ++probes)
	{
		const DevKeyEntry* e = &P.keytab[ slot];
		u32 ev = ldu( &e->event);
		if (ev == event) return e;
		if (ev == 0) return 0;
		slot = (slot+1) & P.keymask;
	}
	return 0;
}

// per-lane variant (divergent probes): stop word log slot of an event, 0 if it has none
__device__ __forceinline__ u32 stopIdxOfLane( KP P, u32 event)
{
	if (!event) return 0;
	u32 slot = keyHash( event) & P.keymask;
	for (u32 probes=0; probes<=P.keymask; ++probes)
	{
		const uint4 eq = ld4( &P.keytab[ slot]);		// {event, listBegin, listCount, stopIdx}
		if (eq.x == event) return eq.w;
		if (eq.x == 0) return 0;
		slot = (slot+1) & P.keymask;
	}
	return 0;
}

// ---------------------------------------------------------------- replayPastEvent (cpp:1272-1334)
__device__ __forceinline__ void replayPastEvent( WSR w, KP P, u32 pastEvent, u32 pastStopIdx, u32 r, u32 range)
{
	if (!pastStopIdx) return;
	const StopLog* L = &STOP[ pastStopIdx-1];
	u32 pastStamp = ldu( &L->timestamp);
	if (!pastStamp) return;
	EvData ld; ldEv( ld, &L->d);
	if (ld.sord + range < w.curpos) return;

	u32 nFollow0 = w.nFollow, nDispose0 = w.nDispose;
	forEachTriggerLastFirst( w, P, r, [&]( u32 t) {
		if (w.err) return;
		const uint4 tq = ldu4( TRIG( t));		// {event, sigval, typevar, link}
		if (tq.x == pastEvent) fireSignal( w, P, r, tq.z & 15u, tq.y, tq.z >> 4, ld);
	});
	// a structure delimiter logged after the replayed event cancels the rule (cpp:1306-1321)
	bool cancelled = false;
	forEachTriggerLastFirst( w, P, r, [&]( u32 t) {
		if (cancelled) return;
		const uint4 tq = ldu4( TRIG( t));
		if ((tq.z & 15u) == SIG_DEL)
		{
			const DevKeyEntry* e = lookupKey( P, tq.x);
			u32 esi = e ? ldu( &e->stopIdx) : 0;
			if (esi)
			{
				u32 ts = ldu( &STOP[ esi-1].timestamp);
				if (ts && ts > pastStamp) cancelled = true;
			}
		}
	});
	if (cancelled) deactivateRule( w, P, r);
	for (u32 di=nDispose0; di<w.nDispose; ++di) deactivateRule( w, P, ldu( &DISPOSE[ di]));
	w.nDispose = nDispose0;
	if (w.nFollow != nFollow0) { w.nFollow = nFollow0; w.err = SPD_ERR_PASTFOLLOW; }
}

// ---------------------------------------------------------------- installProgram (cpp:1168-1270)
__device__ __forceinline__ void installProgram( WSR w, KP P, u32 keyevent, const DevKeyRef* K, const EvData& d)
{
	u32 program = ldu( &K->program);
	const DevProgram* G = &P.programs[ program];
	u32 range = ldu( &G->positionRange);
	if (d.sord + range < w.curpos) return;

	u32 r = allocRule( w, P);
	if (w.err) return;
	HRule* R = &RULES[ r];
	u32 count = ldu( &G->initcount) & 0xFFFFu;		// ActionSlot::count is 16 bit (hpp:98)
	R->value = ldu( &G->initsigval); R->count = count; R->flags = F_ACTIVE; R->start_ordpos = 0;
	R->end_ordpos = 0; R->start_origseg = 0; R->start_origpos = 0; R->program = program;
	R->trigMask = 0; R->dataRef = 0; R->ext = 0; R->owner = 0; R->nInline = 0;
	defineDisposeRule( w, P, d.sord + range, r);
	if (w.err) return;

	u32 tb = ldu( &G->trigBegin), tc = ldu( &G->trigCount);
	u64 keymaskbits = 0;
	u32 nofKey = 0;
	bool hasKey = false;
	u32 blk = r, slot = 0, blkMask = 0;		// trigger slots are filled in installation order, 4 per block
	for (u32 j=0; j<tc; ++j)
	{
		const DevTrigDef* D = &P.trigdefs[ tb+j];
		u32 tev = ldu( &D->event), tflags = ldu( &D->flags);
		u32 sigtype = tflags & 15u;
		bool doInstall;
		if (tev == keyevent)
		{
			if (nofKey >= 32u || j >= 64u) { w.err = SPD_ERR_KEYTRIGGERS; return; }
			keymaskbits |= (1ull << j); ++nofKey;
			bool needs = (sigtype == SIG_ANY && count > 1);		// cpp:1159-1166
			if ((tflags & 0x100u) && !hasKey) { hasKey = true; doInstall = needs; }
			else if (sigtype == SIG_DEL) doInstall = needs;
			else doInstall = true;
		}
		else doInstall = true;
		if (doInstall)
		{
			if (slot == 4)
			{
				// more than 4 triggers: continue in another block chained to this one
				u32 nb = allocRule( w, P);
				if (w.err) return;
				HRule* X = &RULES[ nb];
				X->value = 0; X->count = 0; X->flags = F_EXT; X->start_ordpos = 0;
				X->trigMask = 0; X->dataRef = 0; X->ext = 0; X->owner = r; X->nInline = 0;
				RULES[ blk].trigMask = blkMask; RULES[ blk].ext = nb+1;
				blk = nb; slot = 0; blkMask = 0;
			}
			const u32 t = 4*blk + slot;
			HTrig* T = TRIG( t);
			T->event = tev; T->sigval = ldu( &D->sigval); T->typevar = sigtype | (ldu( &D->variable) << 4);
			addTrigger( w, P, t, tev);
			if (w.err) return;
			blkMask |= 1u << slot; ++slot;
		}
	}
	RULES[ blk].trigMask = blkMask;
	w.nInstalled += 1;

	u32 pastEvent = ldu( &K->pastEvent);
	if (pastEvent)
	{
		w.nAlt += 1;
		replayPastEvent( w, P, pastEvent, ldu( &K->pastStopIdx), r, range);
	}
	if (nofKey && (ldu( &R->flags) & F_ACTIVE))
	{
		while (keymaskbits && !w.err)
		{
			u32 j = (u32)__builtin_ctzll( keymaskbits);
			keymaskbits &= keymaskbits-1;
			const DevTrigDef* D = &P.trigdefs[ tb+j];
			fireSignal( w, P, r, ldu( &D->flags) & 15u, ldu( &D->sigval), ldu( &D->variable), d);
		}
	}
}


// ---------------------------------------------------------------- lane-parallel installation
// installEventPrograms (cpp:1137-1157) for up to 64 programs at a time: lane l instantiates program
// l of the batch.  Everything whose ORDER is observable is placed exactly where the sequential
// loop would have put it -- bucket positions, per-position dispose lists, follow events, results,
// dispose entries all advance in lane order (= program order) through ballots and prefix sums.
// Programs that need the rare machinery (alternative key replay, far expiry heap, more than MAXT
// trigger templates, several key triggers, sub-match data to join) send their whole batch down
// the sequential path, which keeps the order intact.
enum {MAXT=3};


__device__ __forceinline__ void installBatch( WSR w, KP P, u32 keyevent, u32 lb, u32 lc, const EvData& d)
{
	for (u32 base=0; base<lc && !w.err; base+=64)
	{
		const u32 nb = (lc - base) < 64u ? (lc - base) : 64u;
		const bool have = LANE < nb;
		u32 program = 0, pastEvent = 0, pastStopIdx = 0;
		u32 g_initsigval = 0, g_count = 0, g_event = 0, g_handle = 0, g_fmt = 0, g_range = 0, g_tb = 0, g_tc = 0;
		if (have)
		{
			const uint4 kq = ld4( &P.keylist[ lb + base + LANE]);		// {program, pastEvent, pastStopIdx, -}
			program = kq.x; pastEvent = kq.y; pastStopIdx = kq.z;
			const uint4 g0 = ld4( &P.programs[ program]), g1 = ld4( W( &P.programs[ program]) + 4);
			g_initsigval = g0.x; g_count = g0.y & 0xFFFFu; g_event = g0.z; g_handle = g0.w;
			g_fmt = g1.x; g_range = g1.y; g_tb = g1.z; g_tc = g1.w;
		}
		const u32 expiry = d.sord + g_range;
		const bool live = have && !(expiry < w.curpos);			// cpp:1171-1181

		// trigger templates of my program (cpp:1204-1250)
		u32 tEvent[ MAXT], tSigval[ MAXT], tTypevar[ MAXT];
		bool tInstall[ MAXT], tKey[ MAXT];
		u32 nofKey = 0;
		bool hasKey = false;
#pragma unroll
		for (int j=0; j<MAXT; ++j)
		{
			tInstall[ j] = false; tKey[ j] = false; tEvent[ j] = 0; tSigval[ j] = 0; tTypevar[ j] = 0;
			if (live && (u32)j < g_tc)
			{
				const uint4 dq = ld4( &P.trigdefs[ g_tb + j]);			// {event, sigval, variable, flags}
				u32 tev = dq.x, fl = dq.w, sigtype = fl & 15u;
				tEvent[ j] = tev; tSigval[ j] = dq.y; tTypevar[ j] = sigtype | (dq.z << 4);
				bool doInstall;
				if (tev == keyevent)
				{
					tKey[ j] = true; ++nofKey;
					bool needs = (sigtype == SIG_ANY && g_count > 1);
					if ((fl & 0x100u) && !hasKey) { hasKey = true; doInstall = needs; }
					else if (sigtype == SIG_DEL) doInstall = needs;
					else doInstall = true;
				}
				else doInstall = true;
				tInstall[ j] = doInstall;
			}
		}
		// ---- signals on the fresh slot, computed in registers before anything is stored:
		//      (1) alternative-keyed programs replay the logged original key event (cpp:1253-1258 -> :1272-1334),
		//      (2) the key trigger fires (cpp:1259-1269 -> fireSignal cpp:772-979)
		u32 value = g_initsigval, count = g_count, flags = F_ACTIVE, end_ordpos = 0;
		u32 start_ordpos = 0, start_origseg = 0, start_origpos = 0, dataRef = 0;
		bool match = false, fin = false, del = false, odd = false;
		u32 nFires = 0;
		u32 itemVar0 = 0;				// captured variable of the replayed event
		EvData ld; ld.sseg = ld.eseg = ld.spos = ld.epos = ld.sord = ld.eord = ld.sub = ld.fmt = 0;
		auto fireLocal = [&]( u32 sigtype, u32 sigval, const EvData& e, bool& took) {
			bool m = false, f = false; took = false;
			switch (sigtype)
			{
				case SIG_ANY:
					took = true;
					if (count > 0) { m = true; --count; f = (count == 0); if (end_ordpos < e.eord) end_ordpos = e.eord; }
					break;
				case SIG_AND:
					if (count > 0)
					{
						if (!value) { value = e.sord; if (end_ordpos > e.eord) end_ordpos = e.eord; }
						if (value == e.sord) { m = true; --count; f = (count == 0); took = true; }
					}
					break;
				case SIG_SEQUENCE:
				case SIG_SEQUENCE_IMM:
					if (sigval == value && ((sigtype == SIG_SEQUENCE) ? (end_ordpos <= e.sord) : (end_ordpos == e.sord)))
					{
						end_ordpos = e.eord; value = sigval-1;
						if (count > 0) { --count; m = (count == 0); } else m = true;
						f = (value == 0); took = true;
					}
					break;
				case SIG_WITHIN:
					if ((sigval & value) != 0 && end_ordpos <= e.sord)
					{
						end_ordpos = e.eord; value &= ~sigval;
						if (count > 0) { --count; m = (count == 0); } else m = true;
						f = (value == 0); took = true;
					}
					break;
				default:	// SIG_DEL
					count = 0; value = 0; del = true;
					break;
			}
			if (took)
			{
				if (start_ordpos == 0) { start_ordpos = e.sord; start_origseg = e.sseg; start_origpos = e.spos; }
				else if (start_ordpos > e.sord)
				{
					start_ordpos = e.sord;
					if (start_origseg > e.sseg || (start_origseg == e.sseg && start_origpos > e.spos)) { start_origseg = e.sseg; start_origpos = e.spos; }
				}
			}
			if (m) { match = true; if (f) fin = true; }
			++nFires;
		};
		bool hasDel = false;
#pragma unroll
		for (int j=0; j<MAXT; ++j) if (tInstall[ j] && (tTypevar[ j] & 15u) == SIG_DEL) hasDel = true;
		if (live && pastEvent)
		{
			const u32 psi = pastStopIdx;
			if (psi)
			{
				const u32* L = (const u32*)&STOP[ psi-1];
				const uint4 la = ld4( L), lbq = ld4( L+4), lc4 = ld4( L+8);
				if (lc4.x /*timestamp*/ && lbq.x /*start_ordpos*/ + g_range >= w.curpos)
				{
					ld.sseg = la.x; ld.eseg = la.y; ld.spos = la.z; ld.epos = la.w; ld.sord = lbq.x; ld.eord = lbq.y; ld.sub = lbq.z; ld.fmt = lbq.w;
					if (ld.sub) odd = true;				// sub-match data to join: sequential path
					u32 replays = 0;
#pragma unroll
					for (int j=MAXT-1; j>=0; --j)			// the rule's trigger list: last installed first
					{
						if (tInstall[ j] && tEvent[ j] == pastEvent)
						{
							bool took;
							fireLocal( tTypevar[ j] & 15u, tSigval[ j], ld, took);
							if (took && (tTypevar[ j] >> 4)) itemVar0 = tTypevar[ j] >> 4;
							++replays;
						}
					}
					if (replays > 1 || match || del) odd = true;	// a replay that matches or deletes: sequential path
					// a structure delimiter logged after the replayed event cancels the rule (cpp:1306-1321)
					if (hasDel)
					{
#pragma unroll
						for (int j=0; j<MAXT; ++j)
						{
							if (tInstall[ j] && (tTypevar[ j] & 15u) == SIG_DEL)
							{
								const u32 esi = stopIdxOfLane( P, tEvent[ j]);
								if (esi)
								{
									const u32 ts = STOP[ esi-1].timestamp;
									if (ts && ts > lc4.x) odd = true;		// cancelled: sequential path does the deactivation
								}
							}
						}
					}
				}
			}
		}
		u32 keyVar[ MAXT];				// captured variable of key trigger j if it took the event
		bool matchedBare = false;			// matched while the slot had no captured data yet
#pragma unroll
		for (int j=0; j<MAXT; ++j)
		{
			keyVar[ j] = 0;
			if (live && tKey[ j])
			{
				bool took; const bool matchedBefore = match;
				bool hadItem = itemVar0 != 0;
#pragma unroll
				for (int i=0; i<MAXT; ++i) if (i < j && keyVar[ i]) hadItem = true;
				fireLocal( tTypevar[ j] & 15u, tSigval[ j], d, took);
				if (took) keyVar[ j] = tTypevar[ j] >> 4;
				if (match && !matchedBefore && !hadItem && !keyVar[ j]) matchedBare = true;
			}
		}
		if (nofKey > 1 && del) odd = true;		// several key triggers where one deletes: sequential path
		if (!P.withItems) { itemVar0 = 0; for (int j=0; j<MAXT; ++j) keyVar[ j] = 0; }
		u32 nItems = itemVar0 ? 1u : 0u;
#pragma unroll
		for (int j=0; j<MAXT; ++j) if (keyVar[ j]) ++nItems;
		if (matchedBare && nItems) odd = true;		// result staged before the data reference existed: sequential path
		if (del) { match = false; fin = false; }
		const bool slowLane = live && (odd || expiry >= w.curpos + 64u || g_tc > (u32)MAXT);
		const bool liveAll = live;
		const u64 slowMask = __ballot( slowLane);
		// the batch is cut into runs of ordinary programs (handled by all lanes at once) separated by
		// the rare programs that take the sequential path; runs and singles are processed in list order
		for (u32 segStart=0; segStart<nb && !w.err; )
		{
		const u64 slowAhead = slowMask & ~((1ull << segStart) - 1ull);
		const u32 cut = slowAhead ? (u32)__builtin_ctzll( slowAhead) : nb;
		if (cut == segStart)
		{
			installProgram( w, P, keyevent, &P.keylist[ lb+base+cut], d);
			segStart = cut+1;
			continue;
		}
		const bool live = liveAll && LANE >= segStart && LANE < cut;
		segStart = cut;
		const u64 liveMask = __ballot( live);
		if (!liveMask) continue;
		const u32 nlive = (u32)__popcll( liveMask);
		const u32 rank = (u32)__popcll( liveMask & lanesBelow());

		// ---- rule records: pop from the free stack, then bump
		u32 r = 0;
		{
			const u32 fromStack = w.ruleFreeN < nlive ? w.ruleFreeN : nlive;
			const u32 bump = nlive - fromStack;
			if (w.ruleUsed + bump > CAP_RULES) { ARENA_FAIL; return; }
			if (live) r = rank < fromStack ? RULEFREE[ w.ruleFreeN - 1 - rank] : w.ruleUsed + (rank - fromStack);
			w.ruleFreeN -= fromStack; w.ruleUsed += bump;
		}
		// ---- dispose window (cpp:1072-1076): the rules of one expiry position are kept in definition order
		const u32 widx = expiry & 63u;
		{
			u64 todo = liveMask;
			bool full = false;
			const u32 lens = WINDOW[ LANE];			// the 64 list lengths, one read
			u32 myIdx = 0;
			while (todo)					// one round per distinct expiry position in the batch
			{
				const u32 leader = (u32)__builtin_ctzll( todo);
				const u32 wsel = (u32)__builtin_amdgcn_readlane( widx, leader);
				const bool mine = live && widx == wsel;
				const u64 grp = __ballot( mine);
				const u32 cnt = (u32)__builtin_amdgcn_readlane( lens, wsel);
				const u32 ng = (u32)__popcll( grp);
				winReserve( w, P, wsel, cnt, ng);
				if (w.err) { full = true; break; }
				if (mine) myIdx = cnt + (u32)__popcll( grp & lanesBelow());
				WINDOW[ wsel] = cnt + ng;
				todo &= ~grp;
			}
			if (full) return;
			if (live) WINARR[ winEntryIndex( w, P, widx, myIdx)] = r;	// all lists at once
		}
		// ---- triggers: bucket positions in (program, template) order
		u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;		// my installs per bucket, 16 x 8 bit
		u32 hB[ MAXT];
#pragma unroll
		for (int j=0; j<MAXT; ++j)
		{
			hB[ j] = evhash( tEvent[ j]) & 15u;
			if (live && tInstall[ j])
			{
				u32 inc = 1u << ((hB[ j] & 3u)*8);
				u32 ws = hB[ j] >> 2;
				if (ws == 0) c0 += inc; else if (ws == 1) c1 += inc; else if (ws == 2) c2 += inc; else c3 += inc;
			}
		}
		u32 i0 = c0, i1 = c1, i2 = c2, i3 = c3;		// inclusive prefix sums over the lanes (fields stay < 256: 64 x MAXT)
		i0 = waveScanAdd( i0); i1 = waveScanAdd( i1); i2 = waveScanAdd( i2); i3 = waveScanAdd( i3);
		const u32 e0 = i0 - c0, e1 = i1 - c1, e2 = i2 - c2, e3 = i3 - c3;
		const u32 t0 = (u32)__builtin_amdgcn_readlane( i0, 63), t1 = (u32)__builtin_amdgcn_readlane( i1, 63);
		const u32 t2 = (u32)__builtin_amdgcn_readlane( i2, 63), t3 = (u32)__builtin_amdgcn_readlane( i3, 63);
		const u32 totalTrig = byteSum( t0 + t1 + t2 + t3);
		u32 head = 0, local = 0;		// head: occupied trigger slots of my block
		bool overflow = false;
#pragma unroll
		for (int j=0; j<MAXT; ++j)
		{
			if (live && tInstall[ j])
			{
				const u32 h = hB[ j];
				u32 same = 0;
#pragma unroll
				for (int jj=0; jj<j; ++jj) if (tInstall[ jj] && hB[ jj] == h) ++same;	// my own earlier templates
				const u32 pos = BSIZE[ h] + byteField( e0, e1, e2, e3, h) + same;
				const u32 t = 4*r + local;		// slots in installation order (MAXT <= 4)
				if (pos >= CAP_BUCKET) overflow = true;
				else
				{
					*(uint2*)&BKT[ 2*(h*CAP_BUCKET + pos)] = make_uint2( tEvent[ j], t);
					st4( TRIG( t), tEvent[ j], tSigval[ j], tTypevar[ j], (h << 28) | pos);
					head |= 1u << local;
				}
				++local;
			}
		}
		if (__ballot( overflow)) { ARENA_FAIL; return; }
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		if (LANE < 16) BSIZE[ LANE] += byteField( t0, t1, t2, t3, LANE);
		w.nTrig += totalTrig;
		w.nInstalled += nlive;

		w.nSignals += (u32)__popcll( __ballot( live && (nFires & 1u))) + 2u*(u32)__popcll( __ballot( live && (nFires & 2u))) + 4u*(u32)__popcll( __ballot( live && (nFires & 4u)));
		w.nAlt += (u32)__popcll( __ballot( live && pastEvent != 0));
		// captured variables: the replayed event's item first, then the key triggers' in order (LIFO list: last on top)
		const bool emitFollow = live && match && g_event != 0;
		const bool emitResult = live && match && g_handle != 0;
		u32 nInlineOut = 0;
		{
			const bool keepInline = live && nItems == 1u && local <= 2u && !emitFollow && !emitResult;
			const u32 myItems = (live && !keepInline) ? nItems : 0u;
			if (keepInline)
			{
				HRule* R = &RULES[ r];
				if (itemVar0) { st4( &R->trig[ 2], itemVar0, ld.sseg, ld.eseg, ld.spos); st4( &R->trig[ 3], ld.epos, ld.sord, ld.eord, ld.sub); R->_pad[ 0] = ld.fmt; }
#pragma unroll
				for (int j=0; j<MAXT; ++j) if (keyVar[ j]) { st4( &R->trig[ 2], keyVar[ j], d.sseg, d.eseg, d.spos); st4( &R->trig[ 3], d.epos, d.sord, d.eord, d.sub); R->_pad[ 0] = d.fmt; }
				nInlineOut = 1;
			}
			const u64 ib0 = __ballot( myItems & 1u), ib1 = __ballot( myItems & 2u), ib2 = __ballot( myItems & 4u), rmk = __ballot( myItems != 0);
			if (rmk)
			{
				const u32 ni = (u32)__popcll( ib0) + 2u*(u32)__popcll( ib1) + 4u*(u32)__popcll( ib2), nr = (u32)__popcll( rmk);
				const u32 ri = (u32)__popcll( ib0 & lanesBelow()) + 2u*(u32)__popcll( ib1 & lanesBelow()) + 4u*(u32)__popcll( ib2 & lanesBelow());
				const u32 rr = (u32)__popcll( rmk & lanesBelow());
				const u32 itemFromStack = w.itemFreeN < ni ? w.itemFreeN : ni;
				const u32 refFromStack = w.refFreeN < nr ? w.refFreeN : nr;
				if (w.itemUsed + (ni - itemFromStack) > CAP_ITEMS || w.refUsed + (nr - refFromStack) > CAP_REFS) { ARENA_FAIL; return; }
				if (myItems)
				{
					u32 seq = ri, below = 0;
					if (itemVar0)
					{
						const u32 it = seq < itemFromStack ? ITEMFREE[ w.itemFreeN - 1 - seq] : w.itemUsed + (seq - itemFromStack);
						HItem* I = &ITEMS[ it];
						st4( I, itemVar0, 0, 0, 0);
						st4( W( I) + 4, ld.sseg, ld.eseg, ld.spos, ld.epos);
						st4( W( I) + 8, ld.sord, ld.eord, ld.sub, ld.fmt);
						below = it+1; ++seq;
					}
#pragma unroll
					for (int j=0; j<MAXT; ++j)
					{
						if (keyVar[ j])
						{
							const u32 it = seq < itemFromStack ? ITEMFREE[ w.itemFreeN - 1 - seq] : w.itemUsed + (seq - itemFromStack);
							HItem* I = &ITEMS[ it];
							st4( I, keyVar[ j], below, 0, 0);
							st4( W( I) + 4, d.sseg, d.eseg, d.spos, d.epos);
							st4( W( I) + 8, d.sord, d.eord, d.sub, d.fmt);
							below = it+1; ++seq;
						}
					}
					const u32 rf = rr < refFromStack ? REFFREE[ w.refFreeN - 1 - rr] : w.refUsed + (rr - refFromStack);
					REFS[ 2*rf] = below;
					REFS[ 2*rf+1] = 1u + (emitFollow ? 1u : 0u) + (emitResult ? 1u : 0u);	// rule + follow + result (cpp:941-953)
					dataRef = rf+1;
				}
				w.itemFreeN -= itemFromStack; w.itemUsed += ni - itemFromStack;
				w.refFreeN -= refFromStack; w.refUsed += nr - refFromStack;
			}
		}
		if (match) flags |= F_DONE;
		{
			const u64 fm = __ballot( emitFollow);
			if (fm)
			{
				const u32 nf = (u32)__popcll( fm);
				if (w.nFollow + nf > CAP_FOLLOW) { ARENA_FAIL; return; }
				if (emitFollow)
				{
					HFollow* F = &FOLLOW[ w.nFollow + (u32)__popcll( fm & lanesBelow())];
					st4( F, start_origseg, d.eseg, start_origpos, d.epos);
					st4( W( F) + 4, start_ordpos, end_ordpos, dataRef, g_fmt);
					F->event = g_event;
				}
				w.nFollow += nf;
			}
			const u64 rm = __ballot( emitResult);
			if (rm)
			{
				const u32 nr = (u32)__popcll( rm);
				if (w.nStaged + nr > P.arena.maxStaged) { ARENA_FAIL; return; }
				if (emitResult)
				{
					StagedResult* S = &STAGED[ w.nStaged + (u32)__popcll( rm & lanesBelow())];
					st4( S, program, start_ordpos, end_ordpos, start_origseg);
					st4( W( S) + 4, start_origpos, d.eseg, d.epos, dataRef);
				}
				w.nStaged += nr;
			}
			const bool wantDispose = live && (del || (match && fin));
			const u64 dm = __ballot( wantDispose);
			if (dm)
			{
				const u32 nd = (u32)__popcll( dm);
				if (w.nDispose + nd > CAP_DISPOSE) { ARENA_FAIL; return; }
				if (wantDispose) DISPOSE[ w.nDispose + (u32)__popcll( dm & lanesBelow())] = r;
				w.nDispose += nd;
			}
		}
		if (live)
		{
			HRule* R = &RULES[ r];
			st4( R, value, count, flags, start_ordpos);
			st4( W( R) + 4, end_ordpos, start_origseg, start_origpos, program);
			st4( W( R) + 8, head, dataRef, 0, 0);		// {trigMask, dataRef, ext, owner}
			R->nInline = nInlineOut;
		}
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		} // segments
	}
}

// ---------------------------------------------------------------- doTransition (cpp:981-1064)
__device__ __forceinline__ void doTransition( HWS* wsBlock, u32* wsArena, KP P, u32 event, const EvData data)
{
	const WSV w( wsBlock, wsArena);
	w.open += w.nTrig;
	w.nFollow = 1;		// slot 0 of the follow list is the input event itself: it stays in registers

	for (u32 fi=0; fi<w.nFollow && !w.err; ++fi)
	{
		EvData d = data;
		u32 ev = event;
		if (fi) { ldEv( d, &FOLLOW[ fi].d); ev = ldu( &FOLLOW[ fi].event); }
		w.nDispose = 0;
		TRACE2( 6, fi); TRACE2( 7, ev);

		PROF_DECL;
		// fire the triggers waiting for this event: 64 bucket entries per step, one ballot
		if (ev)
		{
			u32 h = evhash( ev) & 15u;
			u32 n = ldu( &BSIZE[ h]);
			const uint2* bk = (const uint2*)BKT + h*CAP_BUCKET;	// {event, trigger id}
			for (u32 base=0; base<n && !w.err; base+=64)
			{
				u32 i = base + LANE;
				uint2 entry = make_uint2( 0, 0);
				if (i < n) entry = bk[ i];
				u64 m = __ballot( i < n && entry.x == ev);
				while (m && !w.err)
				{
					u32 p = (u32)__builtin_ctzll( m);
					m &= m-1;
					const u32 t = (u32)__builtin_amdgcn_readlane( entry.y, p);
					const uint4 tq = ldu4( TRIG( t));	// {event, sigval, typevar, link}; the rule is the block the slot sits in
					fireSignal( w, P, t >> 2, tq.z & 15u, tq.y, tq.z >> 4, d);
				}
			}
		}
		PROF_ADD( 0);
		// install the programs keyed by this event
		TRACE2( 9, 1);
		uint4 eq = make_uint4( 0, 0, 0, 0);			// {event, listBegin, listCount, stopIdx}
		bool e = false;
		if (ev)
		{
			u32 slot = keyHash( ev) & P.keymask;
			for (u32 probes=0; probes<=P.keymask; ++probes)
			{
				eq = ldu4( &P.keytab[ slot]);		// the whole entry with the probe
				if (eq.x == ev) { e = true; break; }
				if (eq.x == 0) break;
				slot = (slot+1) & P.keymask;
			}
		}
		TRACE2( 9, 2);
		u32 stopIdx = 0;
		if (e)
		{
			stopIdx = eq.w;
			u32 lb = eq.y, lc = eq.z;
			TRACE2( 10, lc);
			if (lc >= SPA_L2_BATCH_MIN && !(P.withItems && d.sub)) installBatch( w, P, ev, lb, lc, d);
			else for (u32 k=0; k<lc && !w.err; ++k) { TRACE2( 11, k); installProgram( w, P, ev, &P.keylist[ lb+k], d); }
		}
		TRACE2( 9, 3);
		PROF_ADD( 1);
		// deactivate rules that finished or were deleted
		if (w.nDispose >= SPA_L2_DEACT_MIN) deactivateBatch( w.raw, w.arena, P, DISPOSE, w.nDispose, false, false, true);
		else for (u32 di=0; di<w.nDispose; ++di) deactivateRule( w, P, ldu( &DISPOSE[ di]));
		PROF_ADD( 2);

		if (stopIdx)
		{
			StopLog* L = &STOP[ stopIdx-1];
			stEv( &L->d, d); L->timestamp = ++w.timestamp;
		}
		else if (d.sub && P.withItems)
		{
			disposeRef( w, P, d.sub);
		}
	}
}

// ---------------------------------------------------------------- result items (patternMatcher.cpp:164-190)
// depth-first walk of an item list; items with sub-lists (and no format) are followed by their
// sub-items.  emit==0 counts only.
__device__ __forceinline__ u32 walkItems( WSR w, KP P, u32 ref, u32* out)
{
	u32 n = 0, sp = 0;
	u32 cur = ldu( &REFS[ 2*(ref-1)]);
	for (u32 guard=0;; ++guard)
	{
		if (guard > (1u<<22)) { w.err = SPD_ERR_INTERNAL; break; }
		if (!cur)
		{
			if (!sp) break;
			cur = ldu( &GSTACK[ --sp]);
			continue;
		}
		const HItem* I = &ITEMS[ cur-1];
		EvData d; ldEv( d, &I->d);
		if (out)
		{
			u32* o = out + (u64)n*7;
			o[0] = ldu( &I->variable); o[1] = d.sord; o[2] = d.eord; o[3] = d.sseg; o[4] = d.spos; o[5] = d.eseg; o[6] = d.epos;
		}
		++n;
		cur = ldu( &I->next);
		if (d.sub && !d.fmt)
		{
			if (sp >= P.arena.maxGStack) { ARENA_FAIL; break; }
			GSTACK[ sp++] = cur;
			cur = ldu( &REFS[ 2*(d.sub-1)]);
		}
	}
	return n;
}

// Patterns with format strings (patternMatcher.cpp:172-188): an item that carries a format handle keeps
// its sub-items to itself (they are the arguments of its format string, not items of the result).  The
// output keeps them in place: item i is followed by the `nsub` records of its subtree and
// fmtout[2i..2i+1] = {format handle, nsub}; the host evaluates the format strings from that.
__device__ __forceinline__ u32 walkItemsFormatted( WSR w, KP P, u32 ref, u32* out, u32* fmtout)
{
	u32 n = 0, sp = 0;
	u32 cur = ldu( &REFS[ 2*(ref-1)]);
	for (u32 guard=0;; ++guard)
	{
		if (guard > (1u<<22)) { w.err = SPD_ERR_INTERNAL; break; }
		if (!cur)
		{
			if (!sp) break;
			sp -= 2;
			cur = ldu( &GSTACK[ sp]);
			const u32 owner = ldu( &GSTACK[ sp+1]);		// 1-based index of the formatted item whose subtree ends here
			if (owner && out) fmtout[ 2*(u64)(owner-1)+1] = n - owner;
			continue;
		}
		const HItem* I = &ITEMS[ cur-1];
		EvData d; ldEv( d, &I->d);
		if (out)
		{
			u32* o = out + (u64)n*7;
			o[0] = ldu( &I->variable); o[1] = d.sord; o[2] = d.eord; o[3] = d.sseg; o[4] = d.spos; o[5] = d.eseg; o[6] = d.epos;
			fmtout[ 2*(u64)n] = d.fmt; fmtout[ 2*(u64)n+1] = 0;
		}
		++n;
		cur = ldu( &I->next);
		if (d.sub)
		{
			if (sp+2 > P.arena.maxGStack) { ARENA_FAIL; break; }
			GSTACK[ sp] = cur; GSTACK[ sp+1] = d.fmt ? n : 0u;
			sp += 2;
			cur = ldu( &REFS[ 2*(d.sub-1)]);
		}
	}
	return n;
}

// the same walk done by one lane for its own result (independent loads per lane); sub-lists nest
// rarely and shallowly: a lane that would need more than 4 levels reports `deep` and the document's
// items are written by the sequential walk instead
__device__ __forceinline__ u32 walkItemsLane( WSR w, KP P, u32 ref, u32* out, bool& deep)
{
	u32 n = 0, sp = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	u32 cur = REFS[ 2*(ref-1)];
	for (u32 guard=0;; ++guard)
	{
		if (guard > (1u<<22)) { deep = true; break; }
		if (!cur)
		{
			if (!sp) break;
			--sp;
			cur = sp == 0 ? s0 : sp == 1 ? s1 : sp == 2 ? s2 : s3;
			continue;
		}
		const Item* I = &ITEMS[ cur-1];
		const uint4 a = ld4( I), b = ld4( W( I) + 4), c = ld4( W( I) + 8);	// {variable,next,-,-} {sseg,eseg,spos,epos} {sord,eord,sub,fmt}
		if (out)
		{
			u32* o = out + (u64)n*7;
			o[0] = a.x; o[1] = c.x; o[2] = c.y; o[3] = b.x; o[4] = b.z; o[5] = b.y; o[6] = b.w;
		}
		++n;
		cur = a.y;
		if (c.z && !c.w)
		{
			if (sp >= 4) { deep = true; break; }
			if (sp == 0) s0 = cur; else if (sp == 1) s1 = cur; else if (sp == 2) s2 = cur; else s3 = cur;
			++sp;
			cur = REFS[ 2*(c.z-1)];
		}
	}
	return n;
}

} // anonymous namespace

// ================================================================== kernel
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SPA_L2_WAVES_PER_EU, 8)))
void spa_l2_match_kernel( L2Params kernelArgs)
{
	KP P = kernelParams();
	__shared__ __attribute__((aligned(16))) u32 ldsSlice[ 64 + 16 + 64 + 512 + 16*SCRLDS_CAP + EXPLIST_CAP];
	const WSV w( (HWS*)ldsSlice, P.arenaBase + (u64)blockIdx.x * P.arena.totalWords);
	const u32 ndocs = P.docList ? ldu( P.docListCount) : P.ndocs;
	const u32 waveSlot = blockIdx.x;
	const u32 nWaveSlots = gridDim.x;

	// every wave starts with document `waveSlot` and takes its next ones from a device-side cursor
	// (documents are long sequential jobs of very different length in real corpora); the loop is
	// bounded so that it ends whatever the cursor holds
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
		if (P.docList) doc = ldu( &P.docList[ doc]);
		TRACE( 1, doc);
		// per-document reset (lane-parallel)
		if (LANE < 16) BSIZE[ LANE] = 0;
		WINDOW[ LANE] = 0;
		for (u32 s=LANE; s<P.nofStopWords; s+=64) STOP[ s].timestamp = 0;
		w.curpos = 0; w.timestamp = 0; w.nInstalled = 0; w.nAlt = 0; w.nSignals = 0; w.nTrig = 0; w.open = 0;
		w.ruleFreeN = 0; w.ruleUsed = 0; w.itemFreeN = 0; w.itemUsed = 0;
#if defined(SPA_PROF) || defined(SPA_PROF2)
		w.raw->prof[0] = w.raw->prof[1] = w.raw->prof[2] = w.raw->prof[3] = 0;
#endif
		w.winFreeN = 0; w.winUsed = 0;
		w.refFreeN = 0; w.refUsed = 0; w.heapSize = 0; w.nFollow = 0; w.nDispose = 0; w.nStaged = 0; w.err = 0;

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
		u32 curPosition = 0, nEvents = 0;
		for (u64 lbase=lbeg; lbase<lend && !w.err; lbase+=64)
		{
			// coalesced fetch of up to 64 lexems (16 B each), then replayed one by one
			uint4 lx = make_uint4( 0,0,0,0); u32 seg = 0;
			if (lbase + LANE < lend)
			{
				lx = ((const uint4*)P.lexems)[ lbase + LANE];
				if (P.origseg) seg = P.origseg[ lbase + LANE];
			}
			u32 cnt = (lend - lbase) < 64 ? (u32)(lend - lbase) : 64u;
			for (u32 k=0; k<cnt && !w.err; ++k)
			{
				u32 id = __builtin_amdgcn_readlane( lx.x, k), ordpos = __builtin_amdgcn_readlane( lx.y, k);
				u32 origpos = __builtin_amdgcn_readlane( lx.z, k), origsize = __builtin_amdgcn_readlane( lx.w, k);
				u32 origseg = __builtin_amdgcn_readlane( seg, k);
				// PatternMatcherContext::putInput (patternMatcher.cpp:131-162)
				if (curPosition > ordpos) { w.err = SPD_ERR_ORDER; break; }
				else if (curPosition < ordpos) { curPosition = ordpos; PROF_DECL; setCurrentPos( w, P, ordpos); PROF_ADD( 3); }
				else if (origsize >= 0x7FFFFFFFu || origseg >= 0x7FFFFFFFu || origpos >= 0x7FFFFFFFu) { w.err = SPD_ERR_RANGE; break; }
				if (id >= (1u<<29)) { w.err = SPD_ERR_RANGE; break; }
				EvData d;
				d.sseg = origseg; d.eseg = origseg; d.spos = origpos; d.epos = origpos + origsize;
				d.sord = ordpos; d.eord = ordpos+1; d.sub = 0; d.fmt = 0;
				TRACE2( 2, nEvents); TRACE2( 3, id); TRACE2( 4, ordpos);
				doTransition( w.raw, w.arena, P, id /*TermEvent: type bits 0*/, d);
				TRACE2( 5, nEvents);
				++nEvents;
			}
		}

		// fetchResults: the results are reserved first (their records double as the place where the item
		// counts wait), then the items (one reservation per document)
		TRACE2( 12, w.nStaged); TRACE2( 13, w.err);
		u32 nres = w.err ? 0 : w.nStaged;
		u64 resBase = 0;
		if (nres)
		{
			u64 b = 0;
			if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_RESULTS], (unsigned long long)nres);
			resBase = ((u64)bcast0( (u32)(b >> 32)) << 32) | bcast0( (u32)b);
			if (resBase + nres > P.resultCapacity) { w.err = SPD_ERR_OUTPUT; nres = 0; }
		}
		if (P.withItems && nres)
		{
			// pass 1: every lane walks the item list of its own result and counts
			u32 total = 0;
			bool sequential = P.withFormats != 0;	// format arguments nest: those item lists are walked one by one
			for (u32 base=0; base<nres && !sequential; base+=64)
			{
				const u32 ri = base + LANE;
				u32 n = 0; bool deep = false;
				if (ri < nres)
				{
					const u32 ref = STAGED[ ri].dataRef;
					if (ref) n = walkItemsLane( w, P, ref, 0, deep);
					P.results[ (resBase + ri)*9 + 8] = n;
				}
				if (__ballot( deep)) { sequential = true; break; }
				u32 incl = n;
				incl = waveScanAdd( incl);
				total += (u32)__builtin_amdgcn_readlane( incl, 63);
			}
			if (sequential)
			{
				total = 0;
				for (u32 ri=0; ri<nres && !w.err; ++ri)
				{
					u32 ref = ldu( &STAGED[ ri].dataRef);
					if (ref) total += P.withFormats ? walkItemsFormatted( w, P, ref, 0, 0) : walkItems( w, P, ref, 0);
				}
			}
			u64 itemBase = 0;
			if (w.err) nres = 0;
			else if (total)
			{
				u64 b = 0;
				if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_ITEMS], (unsigned long long)total);
				itemBase = ((u64)bcast0( (u32)(b >> 32)) << 32) | bcast0( (u32)b);
				if (itemBase + total > P.itemCapacity) { w.err = SPD_ERR_OUTPUT; nres = 0; }
			}
			if (nres && sequential)
			{
				// deeply nested item lists: walked one result after the other
				u64 ip = itemBase;
				for (u32 ri=0; ri<nres; ++ri)
				{
					u32 ref = ldu( &STAGED[ ri].dataRef);
					u32 n = !ref ? 0u : P.withFormats ? walkItemsFormatted( w, P, ref, P.items + ip*7, P.itemFormat + ip*2) : walkItems( w, P, ref, P.items + ip*7);
					u32* o = P.results + (resBase + ri)*9;
					o[7] = (u32)ip; o[8] = n;
					ip += n;
				}
			}
			else if (nres)
			{
				// pass 2: positions by prefix sum of the counts, every lane copies its own list
				u64 ip = itemBase;
				for (u32 base=0; base<nres; base+=64)
				{
					const u32 ri = base + LANE;
					u32 n = 0;
					if (ri < nres) n = P.results[ (resBase + ri)*9 + 8];
					u32 incl = n;
					incl = waveScanAdd( incl);
					if (ri < nres)
					{
						const u64 mine = ip + (incl - n);
						P.results[ (resBase + ri)*9 + 7] = (u32)mine;
						const u32 ref = STAGED[ ri].dataRef;
						bool deep = false;
						if (ref) (void)walkItemsLane( w, P, ref, P.items + mine*7, deep);
					}
					ip += (u32)__builtin_amdgcn_readlane( incl, 63);
				}
			}
		}
		if (nres)
		{
			// the 7-tuples: one result per lane, 36-byte records
			for (u32 ri=LANE; ri<nres; ri+=64)
			{
				const StagedResult* S = &STAGED[ ri];
				u32* o = P.results + (resBase + ri)*9;
				const uint4 g0 = ld4( &P.programs[ S->program]);		// {initsigval,initcount,event,resultHandle}
				o[0] = g0.w; o[1] = S->sord; o[2] = S->eord; o[3] = S->sseg; o[4] = S->spos; o[5] = S->eseg; o[6] = S->epos;
				if (P.withFormats) P.resultFormat[ resBase + ri] = P.programs[ S->program].formatHandle;
				if (!P.withItems) { o[7] = 0; o[8] = 0; }
			}
		}
		if (LANE == 0)
		{
			P.docRange[ 2*(u64)doc] = resBase; P.docRange[ 2*(u64)doc+1] = nres;
			u64* st = P.docStats + 4*(u64)doc;
			st[0] = w.nInstalled; st[1] = w.nAlt; st[2] = w.nSignals; st[3] = w.open;
#ifdef SPA_PROF2
			st[0] = w.refUsed; st[1] = w.ruleUsed; st[2] = w.nTrig; st[3] = w.itemUsed;	// high-water marks
#endif
			P.docStatus[ doc] = (int32_t)w.err;
			if (w.err) atomicAdd( (unsigned long long*)&P.counters[ SPC_FAILED], 1ull);
			else atomicAdd( (unsigned long long*)&P.counters[ SPC_EVENTS], (unsigned long long)nEvents);	// (a failed document runs again: its events count then)
#if defined(SPA_PROF) || defined(SPA_PROF2)
			for (int pi=0; pi<4; ++pi) atomicAdd( (unsigned long long*)&P.counters[ 4+pi], (unsigned long long)w.raw->prof[ pi]);
#endif
		}
	}
}

// host-side launcher (called from capi.cpp, same translation unit family compiled by hipcc)
namespace spa {
hipError_t launchL2Match( const L2Params& P, unsigned nblocks /*= waves*/, hipStream_t stream)
{
	hipLaunchKernelGGL( spa_l2_match_kernel, dim3( nblocks), dim3( 64), 0, stream, P);
	return hipGetLastError();
}
}
