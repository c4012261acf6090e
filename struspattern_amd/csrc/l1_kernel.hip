// Level-1 pattern lexer on gfx950: one wavefront per document, one 64-bit automaton word per lane.
//
// What it replaces (reference, CPU): hs_scan (Intel Hyperscan, src/patternLexer.cpp:875-879), the
// per-match callback PatternLexerContext::match_event_handler (:717-826) and the ordinal-position
// pass of PatternLexerContext::match (:893-945).
//
// Two kernels per batch (round 2; one fused kernel before): the SCAN kernel runs stage 1 and leaves the raw
// reports of a document in its slice of a report queue in HBM (16 B per report, ~3 B per text byte); the
// POST kernel runs stages 1b-4 on them.  Fused, the two halves shared one register allocation: 106 scalar
// registers of the byte loop were spilled and every step reloaded kernel arguments.
// Stages per document:
//  1. SCAN   bit-parallel position automaton (tables: l1_tables.h).  Lane l owns word l of every
//            pass; the document byte is wave-uniform, the table row of its byte class is one
//            coalesced 512-byte read; a report is detected with one __ballot per pass and byte.
//            Reports are appended to a queue in (end offset, pattern index) order = the order in
//            which the reference's handler is called.
//  1b. LITERALS  \bWORD\b patterns are no automaton bits: per 64-byte tile the runs of word characters come out
//            of one ballot, their hashes out of one segmented scan, and every lane where a run has ended probes
//            the literal table by itself; the literal reports are merged with the automaton's by (end offset,
//            pattern index).
//  2. SOM    start of match: one lane per queued report runs the pattern's automaton BACKWARDS from
//            the end offset (predecessor sets from the same shift/self/exception tables) and keeps
//            the smallest offset at which a start position is live = leftmost start.
//  3. HANDLER the reference's supersede / ignore / sorted-insert logic, executed in report order on
//            the wave's event array (wave-uniform control flow, scalarised loads).
//  4. ORDPOS ordinal positions and the output records, once per document.
// Integer/byte work, HBM-streaming input: no MFMA.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "l1_tables.h"
#include "l1_device.h"
#include "wave_scan.h"

using namespace spa;

namespace {

typedef uint32_t u32;
typedef uint64_t u64;

#define LANE ((u32)(threadIdx.x & 63u))

#ifdef SPA_PROF
#define PROF_T() __builtin_amdgcn_s_memtime()
#define PROF_ACC( SLOT, T0) do { w.prof[ SLOT] += __builtin_amdgcn_s_memtime() - (T0); } while (0)
#else
#define PROF_T() 0ull
#define PROF_ACC( SLOT, T0) do { (void)(T0); } while (0)
#endif

enum {L1D_OK=0, L1D_ERR_ARENA=2, L1D_ERR_LEXEMSIZE=7, L1D_ERR_INTERNAL=8, L1D_ERR_OUTPUT=9,
      L1D_CHUNK_UNPROVEN=100};	// (internal: the document goes to the sequential re-scan)

__device__ __forceinline__ u32 uni( u32 v) { return __builtin_amdgcn_readfirstlane( v); }
__device__ __forceinline__ u32 ldu( const u32* p) { return __builtin_amdgcn_readfirstlane( *p); }

struct Event { u32 id, origpos, origsize, levelBind; };	// levelBind = level | posbind<<8   (MatchEvent, patternLexer.cpp:665-679)

// hot tables: one image [charMask][acceptMask][startMask][shiftDst][selfLoop][exSrc][exDst], read either
// from the workgroup's LDS copy (ds_read, LDS=true) or from global memory (LDS=false)
extern __shared__ u64 ldsImage[];
template <bool LDS>
struct LexTab
{
	const u64* g;		// global image
	u32 oChar, oAccept, oStart, oShift, oSelf, oExSrc, oExDst;	// (offsets may be biased, modulo 2^32, by the passes an image leaves out: the words kernel's)
	__device__ __forceinline__ u64 at( u32 off) const { if (LDS) return ldsImage[ off]; else return g[ off]; }
};

struct EventLanes { u32 id, pos, size, lb; };	// one event per lane

struct LexWave
{
	u32* queue;		// raw reports: 4 words {to, pattern, accLo, accHi}; after SOM word 2 holds `from`
	Event* events;
	u32 nQueue, nEvents, err, queueCap;
	EventLanes e; u32 cnt;	// the cnt most recent events, lane L holds event nEvents-1-L
	u32 tailPos, tailEnd;	// start and end of the last event (lane 0), valid while cnt > 0
	u32 unit0, unit, unitEnd; u64 docBegin;	// post-processing of a chunked document: the unit whose reports are being read, one behind its last unit
	const unsigned char* doc;
	u32 docLen;
#ifdef SPA_PROF
	u64 prof[4];
#endif
};

// class by code point of the well-formed multi-byte character that begins at `at` (L1Params::cpBlocks), 0xFF = none:
// the byte is classed as a byte then
__device__ __forceinline__ u32 cpClassAt( const L1Params& P, const unsigned char* doc, u32 len, u32 at)
{
	const u32 c = doc[ at];
	const u32 want = (c >= 0xC2u && c <= 0xDFu) ? 2u : (c >= 0xE0u && c <= 0xEFu) ? 3u : (c >= 0xF0u && c <= 0xF4u) ? 4u : 1u;
	if (want == 1u || at + want > len) return 0xFFu;
	u32 v = c & (0xFFu >> (want+1u));
	for (u32 i=1; i<want; ++i)
	{
		const u32 x = doc[ at+i];
		if ((x & 0xC0u) != 0x80u) return 0xFFu;
		v = (v << 6) | (x & 0x3Fu);
	}
	if (want == 2u ? v < 0x80u : want == 3u ? v < 0x800u : (v < 0x10000u || v > 0x10FFFFu)) return 0xFFu;		// overlong / out of range
	return P.cpPages[ (u32)P.cpBlocks[ v >> 6]*64u + (v & 63u)];
}

// class and context of the byte at `pos`: the lead byte of a well-formed character by its code point when the tables
// have such classes; with UCP a continuation byte by whether its character is a word character (twin classes 256..319)
__device__ __forceinline__ void classCtxAt( const L1Params& P, const unsigned char* doc, u32 len, long pos, u32& cls, int& ctx)
{
	if (pos < 0 || pos >= (long)len) { cls = 0; ctx = CTX_EDGE; return; }
	const u32 b = doc[ pos];
	cls = P.byteClass[ b]; ctx = P.classCtx[ cls];
	if (!P.cpBlocks || b < 0x80u) return;
	if (b >= 0xC2u)
	{
		const u32 c = cpClassAt( P, doc, len, (u32)pos);
		if (c != 0xFFu) { cls = c; ctx = P.classCtx[ c]; }
		return;
	}
	if (!P.ucp || b > 0xBFu) return;
	for (long d=1; d<=3 && pos-d >= 0; ++d)
	{
		const u32 l = doc[ pos-d];
		if (l >= 0xC2u && l <= 0xF4u)
		{
			const u32 want = l <= 0xDFu ? 2u : l <= 0xEFu ? 3u : 4u;
			if (want > (u32)d)
			{
				const u32 c = cpClassAt( P, doc, len, (u32)(pos-d));
				if (c != 0xFFu && P.classCtx[ c] == (u32)CTX_WORD) { cls = P.byteClass[ 256u + (b - 0x80u)]; ctx = CTX_WORD; }
			}
			return;
		}
		if ((l & 0xC0u) != 0x80u) return;
	}
}
template <bool CP>
__device__ __forceinline__ int ctxAt( const L1Params& P, const unsigned char* doc, u32 len, long pos)
{
	if (pos < 0 || pos >= (long)len) return CTX_EDGE;
	if (CP && P.ucp) { u32 cls; int ctx; classCtxAt( P, doc, len, pos, cls, ctx); return ctx; }
	return P.classCtx[ P.byteClass[ doc[ pos]]];
}
// UCP: which of the bytes q, q+1, q+2 are continuation bytes of a word character that begins before q (the carry a scan
// starting at q takes over from the tile before)
__device__ __forceinline__ u64 wordCarryAt( const L1Params& P, const unsigned char* doc, u32 len, u32 q)
{
	u64 carry = 0;
	for (u32 d=0; d<3u && q+d<len; ++d)
	{
		const u32 pos = q + d;
		if ((doc[ pos] & 0xC0u) != 0x80u) break;
		for (u32 back=1; back<=3u && back<=pos; ++back)
		{
			const u32 l = doc[ pos-back];
			if (l >= 0xC2u && l <= 0xF4u)
			{
				const u32 want = l <= 0xDFu ? 2u : l <= 0xEFu ? 3u : 4u;
				if (pos-back < q && want > back)
				{
					const u32 c = cpClassAt( P, doc, len, pos-back);
					if (c != 0xFFu && P.classCtx[ c] == (u32)CTX_WORD) carry |= 1ull << d;
				}
				break;
			}
			if ((l & 0xC0u) != 0x80u) break;
		}
	}
	return carry;
}
// class | context << 8 of the byte of my lane in the 64-byte tile at `tile` (`mine`), from the in-register byte tables and
// -- in the kernel instances for tables with classes by code point (CP) -- from the decoded character; wcarry: continuation
// bytes at the start of the next tile that belong to a word character of this one (UCP)
template <bool CP>
__device__ __forceinline__ u32 tileClassCtx( const L1Params& P, const unsigned char* doc, u32 len, u32 tile, u32 mine, u32 clsReg, u32 ctxReg, u64& wcarry)
{
	const u32 shm = (mine & 3u)*8;
	const u32 clsL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((mine >> 2) << 2), (int)clsReg) >> shm) & 0xFFu;
	const u32 ctxL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((mine >> 2) << 2), (int)ctxReg) >> shm) & 0xFFu;
	u32 ccv = clsL | (ctxL << 8);
	if (CP && P.cpBlocks)
	{
		u32 n = 0; bool word = false;
		if (mine >= 0xC2u && mine <= 0xF4u && tile + LANE < len)
		{
			const u32 c = cpClassAt( P, doc, len, tile + LANE);
			if (c != 0xFFu)
			{
				u32 ctx = ctxL;
				if (P.ucp) { ctx = P.classCtx[ c]; word = ctx == (u32)CTX_WORD; n = mine <= 0xDFu ? 2u : mine <= 0xEFu ? 3u : 4u; }
				ccv = c | (ctx << 8);
			}
		}
		if (P.ucp)
		{
			const u64 m2 = __ballot( word && n >= 2u), m3 = __ballot( word && n >= 3u), m4 = __ballot( word && n >= 4u);
			const u64 covered = (m2 << 1) | (m3 << 2) | (m4 << 3) | wcarry;
			wcarry = (m2 >> 63) | (m3 >> 62) | (m4 >> 61);
			if (((covered >> LANE) & 1ull) && mine >= 0x80u && mine <= 0xBFu && tile + LANE < len)
			{
				ccv = (u32)P.byteClass[ 256u + (mine - 0x80u)] | ((u32)CTX_WORD << 8);
			}
		}
	}
	return ccv;
}
// ---------------------------------------------------------------- stage 2: leftmost start per report
// lane-parallel: lane i resolves report base+i
struct LaneReport { u32 to, from, id, levelBind, prefixLen, suffixLen, pi, def, skip; };	// one queued report per lane, pattern attributes attached

// the leftmost start of the matches of one pattern (p0 = {id, word, levelBind, prefixLen}, mask = its bits) that end at `to`, R = the
// positions that consumed byte to-1 and accept there; `to` itself when there is none (a candidate that does not confirm)
template <bool LDS, bool CP>
__device__ __forceinline__ u32 leftmostStart( const unsigned char* doc, u32 docLen, const L1Params& P, const LexTab<LDS>& T, u32 word, u64 mask, u32 to, u64 R)
{
	const u32 pass = word >> 6, ln = word & 63u;
	const u64 shiftDst = T.at( T.oShift + pass*64 + ln), selfLoop = T.at( T.oSelf + pass*64 + ln);
	const u32 nEx = P.exCount[ pass];
	u32 from = to;
	long j = (long)to;			// R = positions that consumed byte j-1
	while (R && j > 0)
	{
		int prevctx = ctxAt<CP>( P, doc, docLen, j-2);
		if (R & T.at( T.oStart + (pass*CTX_COUNT + prevctx)*64 + ln)) from = (u32)(j-1);
		if (j-1 == 0) break;
		u64 Rp = ((R & shiftDst) >> 1) | (R & selfLoop);
		for (u32 e=0; e<nEx; ++e)
		{
			const u32 at = (pass*P.maxExceptions + e)*64 + ln;
			const u64 es = T.at( T.oExSrc + at), ed = T.at( T.oExDst + at);
			Rp |= (R & ed) ? es : 0ull;
		}
		u32 cls = P.byteClass[ doc[ j-2]];
		if (CP && P.cpBlocks && doc[ j-2] >= 0x80u) { int cx; classCtxAt( P, doc, docLen, j-2, cls, cx); }
		R = Rp & mask & T.at( T.oChar + (pass*P.nofClasses + cls)*64 + ln);
		--j;
	}
	return from;
}

template <bool LDS, bool CP>
__device__ __forceinline__ void resolveStarts( const u32* queue, const unsigned char* doc, u32 docLen, const L1Params& P, const LexTab<LDS>& T, u32 base, u32 count, LaneReport& out)
{
	u32 i = base + LANE;
	out.to = 0; out.from = 0; out.id = 0; out.levelBind = 0; out.prefixLen = 0; out.suffixLen = 0; out.pi = 0; out.def = 0; out.skip = 0;
	if (LANE < count)
	{
		const uint4 q = *(const uint4*)(queue + 4*(u64)i);	// {to, pattern, accLo|from, accHi}
		u32 to = q.x, pi = q.y & ~(u32)(L1_LITERAL_FLAG | L1_DEAD_FLAG);
		const uint4 p0 = *(const uint4*)&P.patterns[ pi], p1 = *((const uint4*)&P.patterns[ pi] + 1);	// {id,word,levelBind,prefixLen} {suffixLen,maskLo,maskHi,-}
		out.to = to; out.id = p0.x; out.levelBind = p0.z; out.prefixLen = p0.w; out.suffixLen = p1.x; out.pi = pi; out.def = p1.w;
		if (q.y & L1_DEAD_FLAG) out.skip = 1;				// (words kernel: a candidate that did not confirm)
		if (q.y & L1_LITERAL_FLAG) { out.from = q.z; return; }		// whole-word literal / word shape: start already known
		out.from = leftmostStart<LDS,CP>( doc, docLen, P, T, p0.y, ((u64)p1.z << 32) | p1.y, to, ((u64)q.w << 32) | q.z);
	}
}

// The next batch of up to 64 reports with their starts.  An expression cut into several patterns entries reports
// once per entry: the reports of one expression and one end offset are adjacent; the last of them takes the
// leftmost start of the group, the others are marked to be skipped, and a group is never cut by a batch boundary.
template <bool LDS, bool CP>
__device__ __forceinline__ void nextBatch( const u32* queue, const unsigned char* doc, u32 docLen, const L1Params& P, const LexTab<LDS>& T, u32 nq, u32 qb, u32& qn, LaneReport& lr)
{
	qn = (nq - qb) < 64u ? (nq - qb) : 64u;
	resolveStarts<LDS,CP>( queue, doc, docLen, P, T, qb, qn, lr);
	if (!P.splitPatterns) return;
	if (qb + qn < nq)
	{
		const u32 nto = ldu( queue + 4*(u64)(qb+qn)), npi = ldu( queue + 4*(u64)(qb+qn) + 1) & ~(u32)L1_LITERAL_FLAG;
		const u32 ndef = ldu( &P.patterns[ npi].defIndex);
		const u32 lastTo = (u32)__builtin_amdgcn_readlane( lr.to, qn-1), lastDef = (u32)__builtin_amdgcn_readlane( lr.def, qn-1);
		if (nto == lastTo && ndef == lastDef)
		{
			const u32 g = (u32)__builtin_popcountll( __ballot( LANE < qn && lr.to == lastTo && lr.def == lastDef));
			if (g < qn) qn -= g;
		}
	}
	for (u32 d=1; d<64; d<<=1)
	{
		const u32 pf = (u32)__shfl_up( (int)lr.from, d), pt = (u32)__shfl_up( (int)lr.to, d), pd = (u32)__shfl_up( (int)lr.def, d);
		if (LANE >= d && LANE < qn && pt == lr.to && pd == lr.def && pf < lr.from) lr.from = pf;
	}
	const u32 nt = (u32)__shfl_down( (int)lr.to, 1), nd = (u32)__shfl_down( (int)lr.def, 1);
	lr.skip = (LANE + 1u < qn && nt == lr.to && nd == lr.def) ? 1u : 0u;
}

// ---------------------------------------------------------------- stage 3: the reference's handler
__device__ __forceinline__ void ldEvent( Event& e, const Event* p)
{
	e.id = ldu( &p->id); e.origpos = ldu( &p->origpos); e.origsize = ldu( &p->origsize); e.levelBind = ldu( &p->levelBind);
}
__device__ __forceinline__ void stEvent( Event* p, const Event& e)
{
	p->id = e.id; p->origpos = e.origpos; p->origsize = e.origsize; p->levelBind = e.levelBind;
}

// symbol lookup (PatternTable::symbolId, patternLexer.cpp:312-317): exact text of the match
__device__ u32 lookupSymbol( const unsigned char* doc, const L1Params& P, u32 lexemId, u32 from, u32 len)
{
	u32 h = 2166136261u;
	for (u32 b=0; b<4; ++b) h = symbolHashStep( h, (lexemId >> (8*b)) & 0xFFu);
	for (u32 k=0; k<len; ++k) h = symbolHashStep( h, uni( doc[ from+k]));
	if (!h) h = 1;
	u32 slot = h & P.symbolMask;
	for (u32 probes=0; probes<=P.symbolMask; ++probes)
	{
		const DevSymbol* s = &P.symbols[ slot];
		u32 sh = ldu( &s->hash);
		if (!sh) return 0;
		if (sh == h && ldu( &s->lexemId) == lexemId && ldu( &s->len) == len)
		{
			u32 off = ldu( &s->textOffset);
			bool same = true;
			for (u32 k=0; k<len && same; ++k) same = (uni( P.symbolText[ off+k]) == uni( doc[ from+k]));
			if (same) return ldu( &s->symbolId);
		}
		slot = (slot+1) & P.symbolMask;
	}
	return 0;
}

// The reference's handler (patternLexer.cpp:727-822) edits its event array near the tail almost always.  The wave
// keeps the most recent events in registers, one per lane: lane L holds event nEvents-1-L (L < cnt), the older
// ones (indices < nEvents-cnt) are in the wave's arena in global memory.  The three scans of the handler (delete
// pass, ignore pass, insertion point) become ballots over the lanes, element moves become wave shifts.  A scan
// that would run past the registers into the arena is rare; it goes through handleReportArena on the whole array
// in global memory, a literal restatement of the reference's loops.
__device__ __forceinline__ u32 laneFromAbove( u32 v) { return (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x130, 0xF, 0xF, false); }	// wave_shl:1, lane i <- lane i+1
__device__ __forceinline__ u32 laneFromBelow( u32 v) { return (u32)__builtin_amdgcn_update_dpp( 0, (int)v, 0x138, 0xF, 0xF, false); }	// wave_shr:1, lane i <- lane i-1
__device__ __forceinline__ u64 lowMask( u32 n) { return n ? (~0ull >> (64u - n)) : 0ull; }		// n <= 64

// stage 0: the whole handler; stage 1: insertion only (the delete pass has already removed something)
// returns the new number of events
__device__ __noinline__ u32 handleReportArena( Event* events, u32 n, u32 stage, u32 id, u32 patternid, u32 levelBind, u32 from, u32 to)
{
	struct { Event* events; Event tail; bool tailValid; } w;
	w.events = events; w.tailValid = false;
	const u32 level = levelBind & 0xFFu;
	const bool twin = (patternid != id);
	Event ev; ev.id = id; ev.origpos = from; ev.origsize = to-from; ev.levelBind = levelBind;
	if (n && !w.tailValid) { ldEvent( w.tail, &w.events[ n-1]); w.tailValid = true; }
	if (n == 0)
	{
		stEvent( &w.events[0], ev); n = 1;
		if (twin) { Event t = ev; t.id = patternid; stEvent( &w.events[1], t); n = 2; }
		return n;
	}
	const u32 n0 = n;
	const u32 lastPos = from + (to-from);
	u32 nofDeletes = 0;
	if (stage == 0)
	{
		// delete pass (:757-777): from the back while origpos >= the new origpos
		for (u32 k=n; k>0; --k)
		{
			Event m;
			if (k == n0 && w.tailValid) m = w.tail; else ldEvent( m, &w.events[ k-1]);
			if (!(m.origpos >= ev.origpos)) break;
			u32 mlevel = m.levelBind & 0xFFu;
			if ((ev.id == m.id && m.origpos == ev.origpos && mlevel == level)
			||  (mlevel < level && m.origpos + m.origsize <= lastPos))
			{
				// close the gap: lanes move the tail down by one (ascending order, distinct elements)
				for (u32 t=k-1+LANE; t+1<n; t+=64)
				{
					Event x = w.events[ t+1];
					__builtin_amdgcn_wave_barrier();
					w.events[ t] = x;
				}
				--n; ++nofDeletes;
				w.tailValid = false;
			}
		}
		if (!nofDeletes)
		{
			// ignore pass (:778-792)
			for (u32 k=n; k>0; --k)
			{
				Event m;
				if (k == n0 && w.tailValid) m = w.tail; else ldEvent( m, &w.events[ k-1]);
				if (!(m.origpos + m.origsize >= lastPos)) break;
				if ((m.levelBind & 0xFFu) > level && m.origpos <= ev.origpos) return n;
			}
		}
	}
	// insert (:793-822), literal element moves of the reference (see oracle/l1_oracle.cpp for the
	// two-slot quirk of the symbol twin variant)
	const u32 newSlots = twin ? 2u : 1u;
	Event zero; zero.id = 0; zero.origpos = 0; zero.origsize = 0; zero.levelBind = 0;
	for (u32 s=0; s<newSlots; ++s) stEvent( &w.events[ n+s], zero);
	long prev = (long)n + (long)newSlots - 1, mi = (long)n - 1;
	for (; mi >= 0; prev = mi--)
	{
		Event m; ldEvent( m, &w.events[ mi]);
		if (!(m.origpos > ev.origpos)) break;
		stEvent( &w.events[ prev], m);
	}
	++mi;
	stEvent( &w.events[ mi], ev);
	if (twin) { Event t = ev; t.id = patternid; stEvent( &w.events[ mi+1], t); }
	return n + newSlots;
}

// the lanes [keep, cnt) go to the arena
__device__ __forceinline__ void spillLanes( LexWave& w, u32 keep)
{
	if (LANE >= keep && LANE < w.cnt) *(uint4*)&w.events[ w.nEvents - 1u - LANE] = make_uint4( w.e.id, w.e.pos, w.e.size, w.e.lb);
	w.cnt = keep < w.cnt ? keep : w.cnt;
}
__device__ __forceinline__ void reloadLanes( LexWave& w)
{
	enum {RELOAD=32};
	w.cnt = w.nEvents < (u32)RELOAD ? w.nEvents : (u32)RELOAD;
	uint4 x = make_uint4( 0, 0, 0, 0);
	if (LANE < w.cnt) x = *(const uint4*)&w.events[ w.nEvents - 1u - LANE];
	w.e.id = x.x; w.e.pos = x.y; w.e.size = x.z; w.e.lb = x.w;
}
__device__ __forceinline__ void pushEvent( LexWave& w, u32 at, u32 id, u32 pos, u32 size, u32 lb)
{
	// lanes above `at` take their lower neighbour's event, lane `at` the new one
	const u32 a = laneFromBelow( w.e.id), b = laneFromBelow( w.e.pos), c = laneFromBelow( w.e.size), d = laneFromBelow( w.e.lb);
	const bool up = LANE > at, here = LANE == at;
	w.e.id = up ? a : (here ? id : w.e.id);
	w.e.pos = up ? b : (here ? pos : w.e.pos);
	w.e.size = up ? c : (here ? size : w.e.size);
	w.e.lb = up ? d : (here ? lb : w.e.lb);
	++w.cnt; ++w.nEvents;
	if (at == 0) { w.tailPos = pos; w.tailEnd = pos + size; }
}
__device__ __forceinline__ void readTail( LexWave& w)
{
	w.tailPos = (u32)__builtin_amdgcn_readlane( w.e.pos, 0); w.tailEnd = w.tailPos + (u32)__builtin_amdgcn_readlane( w.e.size, 0);
}

__device__ __forceinline__ void handleReport( LexWave& w, const L1Params& P, u32 id, u32 lb, u32 pre, u32 suf, u32 from, u32 to)
{
	if (to - from >= 65535u) { w.err = L1D_ERR_LEXEMSIZE; return; }			// :727-730
	if (lb & (1u<<17))										// sub expression selection
	{
		if (pre + suf > to - from) return;
		from += pre; to -= suf;
	}
	u32 patternid = id;
	if (lb & (1u<<16))
	{
		const u32 sym = uni( lookupSymbol( w.doc, P, id, from, to-from));	// (a call's result counts as lane-varying unless told otherwise)
		if (sym) patternid = sym;
	}
	const u32 levelBind = lb & 0xFFFFu, level = lb & 0xFFu;
	const bool twin = (patternid != id);
	if (w.nEvents + 2 > P.eventCap)
	{
		// (which of the two working sets was too small is counted apart: the host grows only that one)
		if (LANE == 0) atomicAdd( (unsigned long long*)&P.counters[ L1C_OVER_EVENTS], 1ull);
		w.err = L1D_ERR_ARENA; return;
	}
	if (w.cnt >= 62u) spillLanes( w, 32);
	if (w.nEvents == 0 || (w.cnt && w.tailPos < from && w.tailEnd < to))
	{
		// most reports of a new token: the last event starts and ends before it, all three scans stop at once
		pushEvent( w, 0, id, from, to-from, levelBind);
		if (twin) pushEvent( w, 0, patternid, from, to-from, levelBind);
		return;
	}
	const bool more = w.nEvents > w.cnt;		// older events in the arena
	const u32 lastPos = to;
	u32 stage = 0;
	bool arena = false;
	{
		const u32 cnt = w.cnt;
		const u64 valid = lowMask( cnt);
		const u32 lvlL = w.e.lb & 0xFFu, endL = w.e.pos + w.e.size;
		// delete pass (:757-777): the run of events from the back with origpos >= the new origpos
		const u64 ge = __ballot( w.e.pos >= from) & valid;
		const u32 run = (u32)__builtin_ctzll( ~ge);
		if (run == cnt && more) arena = true;
		else
		{
			u64 del = __ballot( (w.e.id == id && w.e.pos == from && lvlL == level) || (lvlL < level && endL <= lastPos)) & lowMask( run);
			if (del)
			{
				stage = 1;
				while (del)
				{
					// remove the highest lane of the set: the lanes from there on take their upper neighbour's event
					const u32 d = 63u - (u32)__builtin_clzll( del);
					del &= ~(1ull << d);
					const u32 a = laneFromAbove( w.e.id), b = laneFromAbove( w.e.pos), c = laneFromAbove( w.e.size), e = laneFromAbove( w.e.lb);
					const bool dn = LANE >= d;
					w.e.id = dn ? a : w.e.id; w.e.pos = dn ? b : w.e.pos; w.e.size = dn ? c : w.e.size; w.e.lb = dn ? e : w.e.lb;
					--w.cnt; --w.nEvents;
				}
				readTail( w);
			}
			else
			{
				// ignore pass (:778-792): the run of events from the back that end at or behind the new end
				const u64 ge2 = __ballot( endL >= lastPos) & valid;
				const u32 run2 = (u32)__builtin_ctzll( ~ge2);
				if (run2 == cnt && more) arena = true;
				else if (__ballot( lvlL > level && w.e.pos <= from) & lowMask( run2)) return;
			}
		}
	}
	u32 at = 0;
	if (!arena)
	{
		// insertion point (:793-822): behind the events from the back that start to the right of the new one
		const u64 gt = __ballot( w.e.pos > from) & lowMask( w.cnt);
		at = (u32)__builtin_ctzll( ~gt);
		if ((at == w.cnt && more) || (twin && at)) arena = true;
	}
	if (arena)
	{
		spillLanes( w, 0);
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		const u32 n = uni( handleReportArena( w.events, w.nEvents, stage, id, patternid, levelBind, from, to));
		__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront");
		w.nEvents = n;
		reloadLanes( w);
		readTail( w);
		return;
	}
	pushEvent( w, at, id, from, to-from, levelBind);
	if (twin) pushEvent( w, 0, patternid, from, to-from, levelBind);
}

__device__ __forceinline__ u32 ld32u( const unsigned char* p) { u32 v; __builtin_memcpy( &v, p, 4); return v; }
__device__ __forceinline__ uint4 ld128u( const unsigned char* p) { uint4 v; __builtin_memcpy( &v, p, 16); return v; }

// The literals of one 64-byte tile, all lanes at once.  Lane l holds byte tile+l: the runs of word characters of
// the tile come out of one ballot, their hashes out of one segmented scan (polynomial hash, l1_tables.h), and every
// lane where a run has just ended looks its word up in the literal table by itself.  A run that crosses the tile
// boundary is carried (hash, length and start so far).
struct LitCarry { bool in; u32 hash, len, start; };
__device__ __forceinline__ void tileLiterals(  const LexWave& w, const L1Params& P, u32 tile, u32 mine, bool isW, LitCarry& c,
						u64& endMask, u32& litFrom, u32& litBegin, u32& litCount, u32& litPi0, u32& litId0, u32& litLb0)
{
	const u64 wm = __ballot( isW);
	const u64 prevW = (wm << 1) | (c.in ? 1ull : 0ull);
	const u64 startM = wm & ~prevW;
	endMask = ~wm & prevW;				// bit e: the run that ended just before byte tile+e
	const u64 sb = startM & (((1ull << LANE) - 1ull) | (1ull << LANE));
	const int ss = sb ? 63 - (int)__builtin_clzll( sb) : -1;	// start of my run inside the tile (-1: it began in an earlier tile)
	u32 H = isW ? mine + 1u : 0u, M = (u32)L1_LITHASH_MUL;
#pragma unroll
	for (int d=1; d<64; d<<=1)
	{
		const u32 Hs = (u32)__shfl_up( (int)H, d), Ms = (u32)__shfl_up( (int)M, d);
		const bool take = isW && (int)LANE >= d && (int)LANE - d >= ss;
		if (take) { H = Hs * M + H; M = Ms * M; }
	}
	const bool carried = isW && ss < 0;
	const u32 Htot = carried ? c.hash * M + H : H;		// (M = MUL^(lane+1) for a carried run)
	u32 runLen = isW ? (carried ? c.len + LANE + 1u : LANE - (u32)ss + 1u) : 0u;
	if (runLen > 65u) runLen = 65u;
	const u32 from = carried ? c.start : tile + (u32)(ss < 0 ? 0 : ss);
	// the run that ended before me = the values of the lane before me
	u32 pH = (u32)__shfl_up( (int)Htot, 1), pLen = (u32)__shfl_up( (int)runLen, 1), pFrom = (u32)__shfl_up( (int)from, 1);
	if (LANE == 0) { pH = c.hash; pLen = c.len; pFrom = c.start; }
	litFrom = pFrom; litBegin = 0; litCount = 0; litPi0 = 0; litId0 = 0; litLb0 = 0;
	if (((endMask >> LANE) & 1ull) && pLen >= 1u && pLen <= 64u)
	{
		const u32 h = literalHashFinish( pH);
		u32 slot = h & P.literalMask;
		// the word's first 16 bytes, read once beside the first probe (the table entry carries its own first 16)
		uint4 dw = make_uint4( 0, 0, 0, 0);
		if (pFrom + 16u <= w.docLen) dw = ld128u( w.doc + pFrom);
		else
		{
			u32 d[ 4] = {0,0,0,0};
			for (u32 q=0; q<16u && pFrom+q<w.docLen; ++q) d[ q>>2] |= (u32)w.doc[ pFrom + q] << (8*(q&3u));
			dw = make_uint4( d[0], d[1], d[2], d[3]);
		}
		const u32 m0 = pLen >= 4u ? 0xFFFFFFFFu : ((1u << (8*pLen)) - 1u);
		const u32 m1 = pLen >= 8u ? 0xFFFFFFFFu : (pLen > 4u ? ((1u << (8*(pLen-4u))) - 1u) : 0u);
		const u32 m2 = pLen >= 12u ? 0xFFFFFFFFu : (pLen > 8u ? ((1u << (8*(pLen-8u))) - 1u) : 0u);
		const u32 m3 = pLen >= 16u ? 0xFFFFFFFFu : (pLen > 12u ? ((1u << (8*(pLen-12u))) - 1u) : 0u);
		for (u32 probes=0; probes<=P.literalMask; ++probes)
		{
			const uint4* ep = (const uint4*)&P.literals[ slot];
			const uint4 e0 = ep[ 0], e1 = ep[ 1], tx = ep[ 2];	// {hash, len, patBegin, patCount} {pat0, id0, levelBind0, textOffset} {text}
			if (!e0.x) break;
			if (e0.x == h && e0.y == pLen)
			{
				bool same = (((dw.x ^ tx.x) & m0) | ((dw.y ^ tx.y) & m1) | ((dw.z ^ tx.z) & m2) | ((dw.w ^ tx.w) & m3)) == 0;
				for (u32 k=16; k<pLen && same; k+=4)		// (a word beyond 16 bytes: the rest from the text pool)
				{
					const u32 rem = pLen - k;
					u32 a;
					if (pFrom + k + 4u <= w.docLen) a = ld32u( w.doc + pFrom + k);
					else
					{
						a = 0;
						for (u32 q=0; q<rem && q<4u; ++q) a |= (u32)w.doc[ pFrom + k + q] << (8*q);
					}
					const u32 b = ld32u( P.literalText + e1.w + k);
					const u32 mask = rem >= 4u ? 0xFFFFFFFFu : ((1u << (8*rem)) - 1u);
					same = ((a ^ b) & mask) == 0;
				}
				if (same)
				{
					litBegin = e0.z; litCount = e0.w; litPi0 = e1.x; litId0 = e1.y; litLb0 = e1.z;
					break;
				}
			}
			slot = (slot+1) & P.literalMask;
		}
	}
	// carry the run that reaches the end of the tile
	c.in = (wm >> 63) & 1ull;
	if (c.in)
	{
		c.hash = uni( (u32)__shfl( (int)Htot, 63)); c.len = uni( (u32)__shfl( (int)runLen, 63)); c.start = uni( (u32)__shfl( (int)from, 63));
	}
}

// ---------------------------------------------------------------- stage 1: forward scan
// Scans the bytes [segBeg, segEnd) of the document (CH: a chunk of it; else the whole document) and queues the raw reports
// of the end offsets segBeg .. segEnd-1, and of segEnd when that is the document's end.  The state at the start of a
// later chunk comes from a warm-up over the WARM bytes before it: S = the state reached from the empty state with starts
// injected (a subset of the true state), F = the state reached from "every position live" without injection (a superset of
// whatever the true state before the warm-up contributes).  The automaton is linear in its state set, so the true state
// is S | (something inside F): F inside S proves S exact -- the usual case after a few words --, otherwise the document is
// handed to the sequential re-scan (L1D_CHUNK_UNPROVEN).
template <int PASSES, bool LDS, bool CP, bool CH>
__device__ void scanDocument( LexWave& w, const L1Params& P, const LexTab<LDS>& T, const u32 segBeg, const u32 segEnd)
{
	enum {WARM=256};
	u64 state[ PASSES];
#pragma unroll
	for (int p=0; p<PASSES; ++p) state[ p] = 0;
	const u32 len = w.docLen;
	// instances for 1..8 passes run tables of exactly that many passes; the 16 / 32 instances run anything up to it
	const u32 nofPasses = (PASSES <= 8) ? (u32)PASSES : uni( P.nofPasses);
	const u32 nofClasses = uni( P.nofClasses), maxEx = uni( P.maxExceptions);
	int prevctx = CTX_EDGE;
	// byte -> class / context without touching memory: lane l keeps the entries of bytes 4l..4l+3
	u32 clsReg = 0, ctxReg = 0;
	for (u32 k=0; k<4; ++k)
	{
		const u32 c = P.byteClass[ 4*LANE + k];
		clsReg |= c << (8*k);
		ctxReg |= (u32)P.classCtx[ c] << (8*k);
	}
	// per-pass constants in registers
	u64 shiftDst[ PASSES], selfLoop[ PASSES]; u32 nExOf[ PASSES];
#pragma unroll
	for (int p=0; p<PASSES; ++p)
	{
		const bool on = (u32)p < nofPasses;
		shiftDst[ p] = on ? T.at( T.oShift + p*64 + LANE) : 0; selfLoop[ p] = on ? T.at( T.oSelf + p*64 + LANE) : 0;
		nExOf[ p] = on ? uni( P.exCount[ p]) : 0;
	}
	u64 wcarry = 0;
	auto runRange = [&]( auto emitTag, auto injectTag, const u32 from, const u32 to, const bool withEnd)
	{
	constexpr bool EMIT = decltype(emitTag)::value, INJECT = decltype(injectTag)::value;
	for (u32 tile=from; (tile < to || (withEnd && tile == to)) && !w.err; tile+=64)
	{
		// 64 document bytes per load, one per lane; replayed byte by byte through a scalar register
		u32 mine = (tile + LANE < len) ? w.doc[ tile + LANE] : 0u;
		u32 inTile = (to - tile) < 64 ? (to - tile) : 64;	// bytes of the range in this tile
		// class and context of my byte (lane-parallel lookup in the register tables), replayed per byte with one readlane
		const u32 ccv = tileClassCtx<CP>( P, w.doc, len, tile, mine, clsReg, ctxReg, wcarry);
		// one byte step; the virtual step behind the last byte (matches that end with the document) is a
		// separate instance so that the hot one carries no end-of-document conditions
		auto step = [&]( auto atEndTag, const u32 k)
		{
			constexpr bool atEnd = decltype(atEndTag)::value;
			const u32 i = tile + k;
			const u32 cc = atEnd ? ((u32)CTX_EDGE << 8) : (u32)__builtin_amdgcn_readlane( ccv, k);
			const u32 cls = cc & 0xFFu;
			const int ctx = (int)(cc >> 8);
			const u32 groupStart = w.nQueue;	// reports of this end offset start here
			// every table row this byte needs, for all passes, before anything depends on them
			u64 accRow[ PASSES], cmRow[ PASSES], stRow[ PASSES];
#pragma unroll
			for (int p=0; p<PASSES; ++p)
			{
				// the tables end at nofPasses
				accRow[ p] = 0; cmRow[ p] = 0; stRow[ p] = 0;
				if ((u32)p < nofPasses)
				{
					accRow[ p] = T.at( T.oAccept + (p*CTX_COUNT + ctx)*64 + LANE);
					cmRow[ p] = T.at( (p*nofClasses + cls)*64 + LANE);
					stRow[ p] = T.at( T.oStart + (p*CTX_COUNT + prevctx)*64 + LANE);
				}
			}
			u64 acc[ PASSES];
			u64 anyAcc = 0;
#pragma unroll
			for (int p=0; p<PASSES; ++p)
			{
				acc[ p] = 0;
				if ((u32)p >= nofPasses) continue;
				const u64 st = state[ p];
				acc[ p] = st & accRow[ p];			// matches ending before byte i
				anyAcc |= acc[ p];
				if (!atEnd)
				{
					u64 nxt = ((st << 1) & shiftDst[ p]) | (st & selfLoop[ p]) | (INJECT ? stRow[ p] : 0ull);
					const u32 nEx = nExOf[ p];
					for (u32 e=0; e<nEx; ++e)
					{
						const u32 at = (p*maxEx + e)*64 + LANE;
						const u64 es = T.at( T.oExSrc + at), ed = T.at( T.oExDst + at);
						nxt |= (st & es) ? ed : 0ull;
					}
					state[ p] = nxt & cmRow[ p];
				}
			}
			if (EMIT && (__ballot( anyAcc != 0) || (CP && P.nofNullable)))
			{
				
#pragma unroll
				for (int p=0; p<PASSES; ++p)
				{
					const u64 a = acc[ p];
					if (!__ballot( a != 0)) continue;
					// per lane: which patterns packed into my word fired.  patOfBit maps an automaton bit to its
					// pattern; the pattern's mask then retires all of its bits at once (two loads per report).
					const u32 word = p*64 + LANE;
					u32 mycount = 0;
					u32 pi0 = 0, pi1 = 0; u64 m0 = 0, m1 = 0;
					{
						u64 rem = a;
						while (rem)
						{
							const u32 bit = (u32)__builtin_ctzll( rem);
							const u32 pi = P.patOfBit[ word*64 + bit];
							const DevLexPattern* pat = &P.patterns[ pi];
							const uint2 mm = *(const uint2*)&pat->maskLo;
							const u64 m = ((u64)mm.y << 32) | mm.x;
							if (mycount == 0) { pi0 = pi; m0 = m; } else if (mycount == 1) { pi1 = pi; m1 = m; }
							rem &= ~m;
							++mycount;
						}
					}
					// exclusive prefix sum of the counts over the lanes (report order = lane order)
					u32 incl = waveScanAdd( mycount);
					u32 total = (u32)__builtin_amdgcn_readlane( incl, 63);
					u32 at = w.nQueue + incl - mycount;
					if (w.nQueue + total > w.queueCap) { w.err = L1D_ERR_ARENA; break; }
					if (mycount)
					{
						u64 rem = a;
						for (u32 x=0; x<mycount; ++x)
						{
							u32 pi; u64 m;
							if (x == 0) { pi = pi0; m = m0; }
							else if (x == 1) { pi = pi1; m = m1; }
							else
							{
								const u32 bit = (u32)__builtin_ctzll( rem);
								pi = P.patOfBit[ word*64 + bit];
								const uint2 mm = *(const uint2*)&P.patterns[ pi].maskLo;
								m = ((u64)mm.y << 32) | mm.x;
							}
							*(uint4*)(w.queue + 4*(u64)at) = make_uint4( i, pi, (u32)(a & m), (u32)((a & m) >> 32));
							rem &= ~m;
							++at;
						}
					}
					w.nQueue += total;
				}
				if (CP && P.nofNullable)
				{
					// ALLOWEMPTY: an expression that matches the empty string reports (i, i) when its empty path holds between the
					// byte before and this one and nothing longer of it ends here (the leftmost start wins)
					for (u32 k=0; k<P.nofNullable && !w.err; ++k)
					{
						const u32 pi = ldu( &P.nullable[ k].pattern), ok = ldu( &P.nullable[ k].emptyOk);
						if (!((ok >> ((u32)prevctx*CTX_COUNT + (u32)ctx)) & 1u)) continue;
						const DevLexPattern* pat = &P.patterns[ pi];
						const u32 word = ldu( &pat->word);
						const u64 mask = ((u64)ldu( &pat->maskHi) << 32) | ldu( &pat->maskLo);
						u64 a = 0;
#pragma unroll
						for (int p=0; p<PASSES; ++p) if ((word >> 6) == (u32)p) a = acc[ p];
						if (__ballot( (word & 63u) == LANE && (a & mask) != 0)) continue;
						if (w.nQueue + 1u > w.queueCap) { w.err = L1D_ERR_ARENA; break; }
						if (LANE == 0) *(uint4*)(w.queue + 4*(u64)w.nQueue) = make_uint4( i, pi, 0, 0);
						++w.nQueue;
					}
				}
				if (!P.reportsOrdered && !w.err)
				{
					// the patterns are packed by size, not in definition order: bring the reports of this end
					// offset into ascending pattern index (the order the reference's handler sees them in)
					const u32 g = w.nQueue - groupStart;
					if (g > 1)
					{
						__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
						if (g <= 64u)
						{
							uint4 e = make_uint4( 0, 0xFFFFFFFFu, 0, 0);
							if (LANE < g) e = *(const uint4*)(w.queue + 4*(u64)(groupStart + LANE));
							u32 rank = 0;
							for (u32 j=0; j<g; ++j) { const u32 pj = (u32)__builtin_amdgcn_readlane( e.y, j); if (pj < e.y) ++rank; }
							if (LANE < g) *(uint4*)(w.queue + 4*(u64)(groupStart + rank)) = e;
						}
						else
						{
							for (u32 a=groupStart+1; a<w.nQueue; ++a)		// insertion sort, one entry at a time
							{
								const u32* src = w.queue + 4*(u64)a;
								const u32 k0 = ldu( &src[0]), k1 = ldu( &src[1]), k2 = ldu( &src[2]), k3 = ldu( &src[3]);
								u32 b = a;
								while (b > groupStart && ldu( &w.queue[ 4*(u64)(b-1)+1]) > k1)
								{
									u32* dst = w.queue + 4*(u64)b; const u32* s2 = w.queue + 4*(u64)(b-1);
									const u32 m0 = ldu( &s2[0]), m1 = ldu( &s2[1]), m2 = ldu( &s2[2]), m3 = ldu( &s2[3]);
									dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3;
									--b;
								}
								u32* dst = w.queue + 4*(u64)b;
								dst[0] = k0; dst[1] = k1; dst[2] = k2; dst[3] = k3;
							}
						}
						__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
					}
				}
				
			}
			prevctx = ctx;
		};
		for (u32 k=0; k<inTile && !w.err; ++k) step( std::false_type(), k);
		if (withEnd && tile + 64 > to && !w.err) step( std::true_type(), inTile);
	}
	};
	if (CH && segBeg)
	{
		const u32 q = segBeg > (u32)WARM ? segBeg - (u32)WARM : 0u;
		const u64 carry0 = (CP && P.ucp) ? wordCarryAt( P, w.doc, len, q) : 0ull;
		u64 F[ PASSES];
#pragma unroll
		for (int p=0; p<PASSES; ++p) state[ p] = ~0ull;
		wcarry = carry0;
		runRange( std::false_type(), std::false_type(), q, segBeg, false);
#pragma unroll
		for (int p=0; p<PASSES; ++p) { F[ p] = state[ p]; state[ p] = 0; }
		prevctx = q ? ctxAt<CP>( P, w.doc, len, (long)q - 1) : (int)CTX_EDGE;
		wcarry = carry0;
		runRange( std::false_type(), std::true_type(), q, segBeg, false);
		if (q)
		{
			u64 bad = 0;
#pragma unroll
			for (int p=0; p<PASSES; ++p) bad |= F[ p] & ~state[ p];
			if (__ballot( bad != 0)) { w.err = L1D_CHUNK_UNPROVEN; return; }
		}
	}
	if (CH) runRange( std::true_type(), std::true_type(), segBeg, segEnd, segEnd == len);
	else runRange( std::true_type(), std::true_type(), 0u, len, true);		// (the whole document: constants for the plain instance)
}

// the reports of a chunked document lie in one slice of the report queue per chunk: on to the next slice that holds any
__device__ __forceinline__ void sliceOf( LexWave& w, const L1Params& P)
{
	w.queue = P.reportQueue + 4*(((w.docBegin + (u64)(w.unit - w.unit0) * P.chunkBytes) * P.queueMul >> 4) + 64ull*w.unit);
	w.nQueue = ldu( &P.reportCount[ w.unit]);
}
__device__ __forceinline__ bool nextSlice( LexWave& w, const L1Params& P)
{
	while (w.unit + 1u < w.unitEnd)
	{
		++w.unit;
		sliceOf( w, P);
		if (w.nQueue) return true;
	}
	return false;
}

// ---------------------------------------------------------------- stages 1b-3: literals, start of match, handler
// The document's raw reports (scan kernel) and its literal reports (found here, tile by tile) go through the
// reference's handler in the order the reference's callback sees them: ascending end offset, ascending pattern
// index inside one end offset.
template <bool LDS, bool CP, bool CH>
__device__ void postDocument( LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	const u32 len = w.docLen;
	u32 ctxReg = 0, clsReg = 0;		// byte -> context, class: lane l keeps the entries of bytes 4l..4l+3
	for (u32 k=0; k<4; ++k) { const u32 c = P.byteClass[ 4*LANE + k]; clsReg |= c << (8*k); ctxReg |= (u32)P.classCtx[ c] << (8*k); }
	u64 wcarry = 0;
	LitCarry carry; carry.in = false; carry.hash = 0; carry.len = 0; carry.start = 0;
	u32 nq = w.nQueue;
	u32 qi = 0, qb = 0, qn = 0;		// next report; the batch [qb, qb+qn) is resolved in lr
	LaneReport lr;
	lr.to = 0; lr.from = 0; lr.id = 0; lr.levelBind = 0; lr.prefixLen = 0; lr.suffixLen = 0; lr.pi = 0; lr.def = 0; lr.skip = 0;
	if (nq) nextBatch<LDS,CP>( w.queue, w.doc, w.docLen, P, T, nq, 0, qn, lr);
	u32 ahead = (LANE < len) ? w.doc[ LANE] : 0u;		// the next tile's bytes are loaded one tile ahead
	for (u32 tile=0; tile<=len && !w.err; tile+=64)
	{
		const u32 mine = ahead;
		ahead = (tile + 64 + LANE < len) ? w.doc[ tile + 64 + LANE] : 0u;
		const u32 inTile = (len - tile) < 64 ? (len - tile) : 64;
		u64 litEnds = 0; u32 litFrom = 0, litBegin = 0, litCount = 0, litPi0 = 0, litId0 = 0, litLb0 = 0;
		if (P.nofLiterals)
		{
			u32 ctxL;
			if (CP && P.ucp) ctxL = tileClassCtx<true>( P, w.doc, len, tile, mine, clsReg, ctxReg, wcarry) >> 8;
			else { const u32 shm = (mine & 3u)*8; ctxL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((mine >> 2) << 2), (int)ctxReg) >> shm) & 0xFFu; }
			const u64 tLit = PROF_T();
			tileLiterals( w, P, tile, mine, LANE < inTile && ctxL == (u32)CTX_WORD, carry, litEnds, litFrom, litBegin, litCount, litPi0, litId0, litLb0);
			PROF_ACC( 1, tLit);
			litEnds &= __ballot( litCount != 0);
		}
		// every report that ends inside this tile: end offsets tile .. tile+63 (a full tile) or .. len (the last one).
		// Two sorted streams -- the automaton's reports (lanes of lr) and the tile's literals (lanes of lit*, one
		// pattern after the other where a literal defines several) -- merged by (end offset, pattern index).
		const u32 lastTo = (tile + 64 > len) ? len : tile + 63u;
		const u64 NOKEY = ~0ull;
		u32 kL = 64, pb = 0, pc = 0, lj = 0, lfrom = 0, lid = 0, llb = 0;
		u64 keyL = NOKEY;
		auto literalAt = [&]()			// the first literal end of litEnds becomes the current literal
		{
			keyL = NOKEY;
			if (!litEnds) return;
			kL = (u32)__builtin_ctzll( litEnds);
			pb = (u32)__builtin_amdgcn_readlane( litBegin, kL); pc = (u32)__builtin_amdgcn_readlane( litCount, kL);
			lfrom = (u32)__builtin_amdgcn_readlane( litFrom, kL);
			lid = (u32)__builtin_amdgcn_readlane( litId0, kL); llb = (u32)__builtin_amdgcn_readlane( litLb0, kL);
			lj = 0;
			keyL = ((u64)(tile + kL) << 32) | (u32)__builtin_amdgcn_readlane( litPi0, kL);
		};
		literalAt();
		while (!w.err)
		{
			const u32 x = qi - qb;
			u64 keyA = NOKEY;
			if (qi < nq) keyA = ((u64)(u32)__builtin_amdgcn_readlane( lr.to, x) << 32) | (u32)__builtin_amdgcn_readlane( lr.pi, x);
			const bool lit = keyL < keyA;
			const u64 key = lit ? keyL : keyA;
			const u32 toNext = (u32)(key >> 32);
			if (toNext > lastTo) break;		// (also when both streams are at their end)
			const u32 hid = lit ? lid : (u32)__builtin_amdgcn_readlane( lr.id, x);
			const u32 hlb = lit ? llb : (u32)__builtin_amdgcn_readlane( lr.levelBind, x);
			const u32 hpre = lit ? 0u : (u32)__builtin_amdgcn_readlane( lr.prefixLen, x);
			const u32 hsuf = lit ? 0u : (u32)__builtin_amdgcn_readlane( lr.suffixLen, x);
			const u32 hfrom = lit ? lfrom : (u32)__builtin_amdgcn_readlane( lr.from, x);
			const u64 tH = PROF_T();
			if (lit || !__builtin_amdgcn_readlane( lr.skip, x)) handleReport( w, P, hid, hlb, hpre, hsuf, hfrom, toNext);
			PROF_ACC( 3, tH);
			if (lit)
			{
				++lj;
				if (lj < pc)
				{
					const u32 lpi = ldu( &P.litPats[ pb + lj]);
					const DevLexPattern* pat = &P.patterns[ lpi];
					lid = ldu( &pat->id); llb = ldu( &pat->levelBind);
					keyL = ((u64)toNext << 32) | lpi;
				}
				else { litEnds &= litEnds - 1; literalAt(); }
			}
			else
			{
				++qi;
				if (qi == qb + qn && qi < nq)
				{
					qb = qi;
					const u64 tR = PROF_T();
					nextBatch<LDS,CP>( w.queue, w.doc, w.docLen, P, T, nq, qb, qn, lr);
					PROF_ACC( 2, tR);
				}
				else if (CH && qi == nq && nextSlice( w, P))
				{
					// (a chunked document: the reports of its next chunk)
					nq = w.nQueue; qi = 0; qb = 0;
					nextBatch<LDS,CP>( w.queue, w.doc, w.docLen, P, T, nq, qb, qn, lr);
				}
			}
		}
	}
	if (!w.err && qi != nq) w.err = L1D_ERR_INTERNAL;
}

__device__ __forceinline__ u64 queueBase( const L1Params& P, u64 bytePos, u32 unit) { return ((bytePos * P.queueMul) >> 4) + 64ull*unit; }
__device__ __forceinline__ void docBounds( const L1Params& P, u32 doc, u64& beg, u64& end);
template <bool LDS> __device__ __forceinline__ void stageTables( const L1Params& P, LexTab<LDS>& T);
// ---------------------------------------------------------------- scan, a lane per stream (round 3)
// When what is left to scan fits a few automaton words (the literals and the word shapes are found by the words kernel: of the
// 10 001 expressions of the benchmark set five are left, 17 positions), a wave that steps through ONE byte per instruction
// stream wastes its lanes.  Here the unit is cut into 64 pieces and every lane runs the whole (small) automaton over its own
// piece: 64 bytes per step of the wave.  The state at the start of a piece comes from the same warm-up proof the chunks of a
// long document use (F inside S, scanDocument); a lane whose proof fails sends the document to the sequential re-scan.  The
// reports of a lane go to the lane's part of the unit's queue slice and are moved together at the end (lane order = offset order).
__shared__ unsigned short laneCc[ 256];		// byte -> class | context << 8
#ifndef SPA_L1_LANE_PARK
#define SPA_L1_LANE_PARK 4
#endif
enum {LANE_PARK=SPA_L1_LANE_PARK};		// 16-byte blocks of a lane's piece parked per refill (8 = a whole 128-byte line: 33 KB per workgroup, 8 waves per CU beside the table image; 4: 16 waves)
__shared__ uint4 laneText[ 4][ LANE_PARK*64];		// per wave: the next bytes of every lane's piece of the text
template <int W>
__device__ __forceinline__ void scanUnitLanes( LexWave& w, const L1Params& P, const LexTab<true>& T, const u32 segBeg, const u32 segEnd)
{
#ifndef SPA_L1_LANE_WARM
#define SPA_L1_LANE_WARM 256
#endif
	enum {WARM=SPA_L1_LANE_WARM};
	const u32 len = w.docLen;
	const u32 nofClasses = uni( P.nofClasses), maxEx = uni( P.maxExceptions), nEx = uni( P.exCount[ 0]);
	const u32 span = segEnd - segBeg;
	const u32 per = (((span + 63u) >> 6) + 15u) & ~15u;		// bytes per lane, a multiple of 16
	u32 b0 = segBeg + LANE*per; if (b0 > segEnd) b0 = segEnd;
	u32 b1 = b0 + per; if (b1 > segEnd) b1 = segEnd;
	u64 shiftDst[ W], selfLoop[ W];
#pragma unroll
	for (int x=0; x<W; ++x) { shiftDst[ x] = T.at( T.oShift + x); selfLoop[ x] = T.at( T.oSelf + x); }
	auto stepWords = [&]( u64* st, u32 cls, u32 prevctx, bool inject)
	{
#pragma unroll
		for (int x=0; x<W; ++x)
		{
			const u64 s0 = st[ x];
			u64 nxt = ((s0 << 1) & shiftDst[ x]) | (s0 & selfLoop[ x]) | (inject ? T.at( T.oStart + prevctx*64 + x) : 0ull);
			for (u32 e=0; e<nEx; ++e)
			{
				const u64 es = T.at( T.oExSrc + e*64 + x), ed = T.at( T.oExDst + e*64 + x);
				nxt |= (s0 & es) ? ed : 0ull;
			}
			st[ x] = nxt & T.at( cls*64 + x);
		}
	};
	u64 S[ W];
#pragma unroll
	for (int x=0; x<W; ++x) S[ x] = 0;
	u32 prevctx = (u32)CTX_EDGE;
	bool unproven = false;
	if (b0 < b1 && b0 > 0)
	{
		// warm-up over the bytes before my piece: F from "every position live" without starts, S from nothing with starts
		const u32 q = b0 > (u32)WARM ? b0 - (u32)WARM : 0u;
		u64 F[ W];
#pragma unroll
		for (int x=0; x<W; ++x) F[ x] = ~0ull;
		prevctx = q ? ((u32)laneCc[ w.doc[ q-1]] >> 8) : (u32)CTX_EDGE;
		// (16 bytes per load: a byte per load was 256 dependent global loads per lane)
#pragma unroll 1
		for (u32 i=q; i<b0; i+=16)
		{
			uint4 v = make_uint4( 0,0,0,0);
			if (i + 16u <= len) v = ld128u( w.doc + i);
			else
			{
				// (the document's last bytes: one at a time through the lane's parking place, nothing is read behind the text)
				uint4* pk = laneText[ threadIdx.x >> 6] + LANE;
				pk[ 0] = make_uint4( 0,0,0,0);
				for (u32 k=0; k<16u && i+k<len; ++k) ((unsigned char*)pk)[ k] = w.doc[ i+k];
				v = pk[ 0];
			}
#pragma unroll 1
			for (u32 dw=0; dw<4u; ++dw)
			{
				const u32 x = dw == 0 ? v.x : (dw == 1 ? v.y : (dw == 2 ? v.z : v.w));
#pragma unroll
				for (int k=0; k<4; ++k)
				{
					if (i + 4u*dw + (u32)k < b0)
					{
						const u32 cc = laneCc[ (x >> (8*k)) & 0xFFu];
						stepWords( F, cc & 0xFFu, prevctx, false);
						stepWords( S, cc & 0xFFu, prevctx, true);
						prevctx = cc >> 8;
					}
				}
			}
		}
		if (q)
		{
			u64 bad = 0;
#pragma unroll
			for (int x=0; x<W; ++x) bad |= F[ x] & ~S[ x];
			unproven = bad != 0;
		}
	}
	if (__ballot( unproven)) { w.err = L1D_CHUNK_UNPROVEN; return; }
	// my part of the unit's queue slice
	const u32 regionCap = w.queueCap >> 6;
	u32* region = w.queue + 4*(u64)(LANE*regionCap);
	u32 cnt = 0; bool full = false;
	auto emit = [&]( u32 to, const u64* st, u32 ctx)
	{
#pragma unroll
		for (int x=0; x<W; ++x)
		{
			u64 rem = st[ x] & T.at( T.oAccept + ctx*64 + x);
			while (rem)
			{
				const u32 bit = (u32)__builtin_ctzll( rem);
				const u32 pi = P.patOfBit[ (u32)x*64 + bit];
				const uint2 mm = *(const uint2*)&P.patterns[ pi].maskLo;
				const u64 m = ((u64)mm.y << 32) | mm.x;
				if (cnt < regionCap) { *(uint4*)(region + 4*(u64)cnt) = make_uint4( to, pi, (u32)(rem & m), (u32)((rem & m) >> 32)); ++cnt; } else full = true;
				rem &= ~m;
			}
		}
	};
	// 64 bytes of my piece per refill, parked in LDS (a lane's loads are 1 KB apart from its neighbours': read 16 bytes at
	// a time straight from memory, every 128-byte line of the text came in eight times -- 11 GB of traffic for 0.8 GB of text;
	// a whole line per refill is 33 KB of LDS per workgroup and 8 waves per CU: 5.7 ms instead of 4.2 on 805 MB)
	uint4* park = laneText[ threadIdx.x >> 6] + LANE;		// [LANE_PARK][64]: block q of lane l at park[ q*64]
	for (u32 i=b0; i<b1; i+=16u*(u32)LANE_PARK)
	{
#pragma unroll 1
		for (u32 q=0; q<(u32)LANE_PARK; ++q)
		{
			const u32 at = i + 16u*q;
			if (at >= b1) break;
			if (at + 16u <= len) park[ q*64] = ld128u( w.doc + at);
			else
			{
				// (the document's last bytes: one at a time, nothing is read behind the text)
				park[ q*64] = make_uint4( 0,0,0,0);
				for (u32 k=0; k<16u && at+k<len; ++k) ((unsigned char*)&park[ q*64])[ k] = w.doc[ at+k];
			}
		}
#pragma unroll 1
		for (u32 q=0; q<(u32)LANE_PARK && i + 16u*q < b1; ++q)
		{
			const uint4 v = park[ q*64];
			const u32 vv[ 4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int k=0; k<16; ++k)
			{
				const u32 at = i + 16u*q + (u32)k;
				if (at < b1)
				{
					const u32 cc = laneCc[ (vv[ k>>2] >> (8*(k&3))) & 0xFFu];
					emit( at, S, cc >> 8);			// matches that end before this byte
					stepWords( S, cc & 0xFFu, prevctx, true);
					prevctx = cc >> 8;
				}
			}
		}
	}
	// matches that end with the document: by the lane that holds its last byte (an empty document has none)
	if (segEnd == len && b1 == segEnd && b0 < b1) emit( len, S, (u32)CTX_EDGE);
	if (__ballot( full)) { w.err = L1D_ERR_ARENA; return; }
	// ---- the lanes' parts moved together (ascending lanes: a part only moves towards the front)
	const u32 incl = waveScanAdd( cnt);
	const u32 excl = incl - cnt;
	__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
	for (u32 l=1; l<64; ++l)
	{
		const u32 n = (u32)__builtin_amdgcn_readlane( cnt, l);
		if (!n) continue;
		const u32 dst = (u32)__builtin_amdgcn_readlane( excl, l), src = l*regionCap;
		if (dst == src) continue;
		for (u32 k=0; k<n; k+=64)
		{
			uint4 rec = make_uint4( 0,0,0,0);
			if (k + LANE < n) rec = *(const uint4*)(w.queue + 4*(u64)(src + k + LANE));
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
			if (k + LANE < n) *(uint4*)(w.queue + 4*(u64)(dst + k + LANE)) = rec;
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
		}
	}
	w.nQueue = (u32)__builtin_amdgcn_readlane( incl, 63);
}

template <int W>
__device__ __forceinline__ void scanDocumentsLanes( const L1Params& P)
{
	LexTab<true> T;
	for (u32 k=threadIdx.x; k<256u; k+=blockDim.x) { const u32 cl = P.byteClass[ k]; laneCc[ k] = (unsigned short)(cl | ((u32)P.classCtx[ cl] << 8)); }
	stageTables( P, T);			// (the barrier inside covers laneCc too)
	const u32 chunked = ldu( (const u32*)&P.counters[ L1C_CHUNKED]);
	LexWave w;
	w.events = 0; w.nEvents = 0; w.cnt = 0; w.tailPos = 0; w.tailEnd = 0;
	const u32 nunits = ldu( (const u32*)&P.counters[ L1C_UNITS]);
	for (u32 round=0; round<=nunits; ++round)
	{
		u32 unit = 0;
		if (LANE == 0) unit = atomicAdd( (u32*)&P.counters[ L1C_CURSOR], 1u);
		unit = uni( unit);
		if (unit >= nunits) break;
		u32 doc = unit;
		if (chunked)
		{
			u32 lo = 0, hi = P.ndocs;			// last document whose first unit is <= unit
			while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (ldu( &P.unitStart[ mid]) <= unit) lo = mid; else hi = mid; }
			doc = lo;
		}
		u64 beg, end;
		docBounds( P, doc, beg, end);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		u32 segBeg = 0, segEnd = w.docLen;
		if (chunked)
		{
			segBeg = (unit - ldu( &P.unitStart[ doc])) * P.chunkBytes;
			segEnd = (w.docLen - segBeg) < P.chunkBytes ? w.docLen : segBeg + P.chunkBytes;
		}
		const u64 qb = queueBase( P, beg + segBeg, unit);
		w.queue = P.reportQueue + 4*qb;
		w.queueCap = (u32)(queueBase( P, beg + segEnd, unit + 1) - qb);
		w.nQueue = 0; w.err = 0;
		scanUnitLanes<W>( w, P, T, segBeg, segEnd);
		if (LANE == 0)
		{
			if (w.err == L1D_CHUNK_UNPROVEN) { P.docSequential[ doc] = 1; P.reportCount[ unit] = 0; }
			else
			{
				P.reportCount[ unit] = w.err ? 0u : w.nQueue;
				if (w.err) P.docStatus[ doc] = (int32_t)w.err;
				if (w.err == L1D_ERR_ARENA) atomicAdd( (unsigned long long*)&P.counters[ L1C_OVER_QUEUE], 1ull);
				atomicAdd( (unsigned long long*)&P.counters[ L1C_RAW], (unsigned long long)w.nQueue);
			}
		}
	}
}

// ---------------------------------------------------------------- words kernel: whole-word literals and word shapes
// A lane per byte, a wave per scan unit (a document, or a chunk of a long one): everything here is parallel over the text.
// Where a run of word characters ends, the lane behind it knows the run (start, length, polynomial hash -- one ballot and one
// segmented scan per 64-byte tile, as tileLiterals) and the run before it, and derives the candidates of that end offset:
// the literal table's entry of the whole word, and per shape variant of the table (l1_tables.h: PREFIX / SUFFIX / PREVWORD) the
// entry of the few bytes that pin the shape.  A shape candidate is confirmed by the backward walk of its automaton from the end
// offset (leftmostStart: exactly what an automaton report of that expression would go through), which also gives the leftmost
// start.  The unit's records go to its slice of the word queue in (end offset, pattern) order; a candidate that does not confirm
// keeps its record, marked dead (the lanes reserve their records before they walk).
struct WordCarry
{
	bool in; u32 hash, len, start;			// the run that reaches the end of the tile
	bool lastValid; u32 lastTo, lastHash, lastLen;	// the last run that ended so far: its end offset (= the byte behind it), hash, length
};

// start of the run of word characters that byte pos-1 belongs to (pos >= 1, byte pos-1 is a word character)
__device__ __forceinline__ u32 runStartBefore( const L1Params& P, const unsigned char* doc, u32 pos, u32 ctxReg)
{
	for (;;)
	{
		const u32 base = pos >= 64u ? pos - 64u : 0u;
		const u32 at = base + LANE;
		const u32 b = at < pos ? (u32)doc[ at] : 0u;
		const u32 ctxL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((b >> 2) << 2), (int)ctxReg) >> ((b & 3u)*8)) & 0xFFu;
		const u64 nonWord = __ballot( at < pos && ctxL != (u32)CTX_WORD);
		if (nonWord) return base + 64u - (u32)__builtin_clzll( nonWord);		// behind the last byte that is none
		if (base == 0) return 0;
		pos = base;
	}
}

// The backward walk of leftmostStart for a candidate of the words kernel: class and context of the bytes it visits come from the
// wave's ring in LDS (the last 128 bytes of the text, written a tile at a time), the automaton's rows from the LDS image -- two
// LDS round trips per byte instead of four dependent global loads.  A walk that leaves the ring goes on with leftmostStart.
enum {WORD_RING=512, WORD_ENDS=96, WORD_ENDWORDS=5};
// (per wave: a ring of class | context << 8 of the last 512 bytes, wordRing, and wordEnds -- both declared in wordsDocuments, one pair
//  of arrays per workgroup size)
// the run ends that wait for their probes, {end offset, start of the run, hash of the run, hash of the word before it, length | length of
// the word before << 8} each; while a batch of them is worked on (they are in registers then) the first 1 KB holds the candidates
// {end offset, pattern | L1_LITERAL_FLAG for a literal, start of the run, -}: a lane each for the walks
static_assert( WORD_RING*2 + WORD_ENDS*WORD_ENDWORDS*4 == L1_WORDS_LDS_PER_WAVE, "static LDS of the words kernel");
static_assert( 64*WORD_ENDWORDS*4 >= 64*16, "the candidates lie over the ends of the batch");
template <bool LDS>
__device__ __forceinline__ u32 confirmWalk( const unsigned char* doc, u32 docLen, const L1Params& P, const LexTab<LDS>& T, const unsigned short* ring, u32 ringLo,
					    u32 word, u64 mask, u32 to)
{
	// ring[ q & 127] = class | context << 8 of byte q for ringLo <= q < ringLo + 128
	const u32 pass = word >> 6, ln = word & 63u;
	const u32 ccLast = ring[ (to-1u) & (WORD_RING-1)];
	const u32 nextctx = to < docLen ? ((u32)ring[ to & (WORD_RING-1)] >> 8) : (u32)CTX_EDGE;
	u64 R = mask & T.at( T.oAccept + (pass*CTX_COUNT + nextctx)*64 + ln) & T.at( T.oChar + (pass*P.nofClasses + (ccLast & 0xFFu))*64 + ln);
	if (!R) return to;
	const u64 shiftDst = T.at( T.oShift + pass*64 + ln), selfLoop = T.at( T.oSelf + pass*64 + ln);
	const u32 nEx = P.exCount[ pass];
	u32 from = to;
	u32 j = to;				// R = positions that consumed byte j-1
	while (R && j > 0)
	{
		if (j >= 2u && j-2u < ringLo)
		{
			// (rare: a match longer than the ring) the rest of the walk from global memory
			const u32 f = leftmostStart<LDS,false>( doc, docLen, P, T, word, mask, j, R);
			return f < j ? f : from;
		}
		const u32 cc = j >= 2u ? (u32)ring[ (j-2u) & (WORD_RING-1)] : ((u32)CTX_EDGE << 8);
		if (R & T.at( T.oStart + (pass*CTX_COUNT + (cc >> 8))*64 + ln)) from = j-1u;
		if (j == 1u) break;
		u64 Rp = ((R & shiftDst) >> 1) | (R & selfLoop);
		for (u32 e=0; e<nEx; ++e)
		{
			const u32 at = (pass*P.maxExceptions + e)*64 + ln;
			const u64 es = T.at( T.oExSrc + at), ed = T.at( T.oExDst + at);
			Rp |= (R & ed) ? es : 0ull;
		}
		R = Rp & mask & T.at( T.oChar + (pass*P.nofClasses + (cc & 0xFFu))*64 + ln);
		--j;
	}
	return from;
}

// One batch of up to 64 run ends, a lane each: literal probe, shape probes, the candidates ordered by (end offset, pattern) into the
// unit's queue, every candidate that is not a literal confirmed by a backward walk on a lane of its own (64 candidates a round).
template <bool LDS>
__device__ __forceinline__ void wordsFlush( LexWave& w, const L1Params& P, const LexTab<LDS>& T, const unsigned short* ring, const u32 ringLo, u32* ends, const u32 count, const u32 nVar)
{
	const u32 len = w.docLen;
	const bool emit = LANE < count;
	u32 to = 0, pFrom = 0, pH = 0, qH = 0, pLen = 0, qLen = 0;
	if (emit)
	{
		const u32* e = ends + WORD_ENDWORDS*LANE;
		to = e[ 0]; pFrom = e[ 1]; pH = e[ 2]; qH = e[ 3];
		const u32 l = e[ 4]; pLen = l & 0xFFu; qLen = l >> 8;
	}
	__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");		// (the candidates overwrite the batch from here on)
	// ---- candidates of my end offset: lists of patterns, ascending each
	enum {NLIST=SHAPE_MAXVARIANTS+1};
	u32 lb[ NLIST], lc[ NLIST];
#pragma unroll
	for (int k=0; k<NLIST; ++k) { lb[ k] = 0; lc[ k] = 0; }
	u32 total = 0, lit0 = 0;
	if (emit)
	{
		if (P.nofLiterals && pLen <= 64u)
		{
			// whole-word literal (the probe of tileLiterals)
			const u32 h = literalHashFinish( pH);
			u32 slot = h & P.literalMask;
			uint4 dw = make_uint4( 0, 0, 0, 0);
			if (pFrom + 16u <= len) dw = ld128u( w.doc + pFrom);
			else
			{
				u32 d[ 4] = {0,0,0,0};
				for (u32 q=0; q<16u && pFrom+q<len; ++q) d[ q>>2] |= (u32)w.doc[ pFrom + q] << (8*(q&3u));
				dw = make_uint4( d[0], d[1], d[2], d[3]);
			}
			const u32 m0 = pLen >= 4u ? 0xFFFFFFFFu : ((1u << (8*pLen)) - 1u);
			const u32 m1 = pLen >= 8u ? 0xFFFFFFFFu : (pLen > 4u ? ((1u << (8*(pLen-4u))) - 1u) : 0u);
			const u32 m2 = pLen >= 12u ? 0xFFFFFFFFu : (pLen > 8u ? ((1u << (8*(pLen-8u))) - 1u) : 0u);
			const u32 m3 = pLen >= 16u ? 0xFFFFFFFFu : (pLen > 12u ? ((1u << (8*(pLen-12u))) - 1u) : 0u);
			for (u32 probes=0; probes<=P.literalMask; ++probes)
			{
				const uint4* ep = (const uint4*)&P.literals[ slot];
				const uint4 e0 = ep[ 0], e1 = ep[ 1], tx = ep[ 2];
				if (!e0.x) break;
				if (e0.x == h && e0.y == pLen)
				{
					bool same = (((dw.x ^ tx.x) & m0) | ((dw.y ^ tx.y) & m1) | ((dw.z ^ tx.z) & m2) | ((dw.w ^ tx.w) & m3)) == 0;
					for (u32 k=16; k<pLen && same; k+=4)
					{
						const u32 rem = pLen - k;
						u32 a;
						if (pFrom + k + 4u <= len) a = ld32u( w.doc + pFrom + k);
						else { a = 0; for (u32 q=0; q<rem && q<4u; ++q) a |= (u32)w.doc[ pFrom + k + q] << (8*q); }
						const u32 b = ld32u( P.literalText + e1.w + k);
						const u32 mask = rem >= 4u ? 0xFFFFFFFFu : ((1u << (8*rem)) - 1u);
						same = ((a ^ b) & mask) == 0;
					}
					if (same) { lb[ 0] = e0.z; lc[ 0] = e0.w; lit0 = e1.x; break; }
				}
				slot = (slot+1) & P.literalMask;
			}
		}
	}
	// ---- shape variants: the key bytes come out of two unaligned words of the text (the run's first four bytes, its last
	// four); the compact table (fingerprint, pattern or list) sits in LDS beside the automaton tables: no global round trip
	if (nVar)
	{
		u32 first4 = 0, last4 = 0;
		if (emit)
		{
			if (pFrom + 4u <= len) first4 = ld32u( w.doc + pFrom); else for (u32 i=0; i<4u && pFrom+i<len; ++i) first4 |= (u32)w.doc[ pFrom+i] << (8*i);
			if (to >= 4u) last4 = ld32u( w.doc + to - 4u); else for (u32 i=0; i<to; ++i) last4 |= (u32)w.doc[ i] << (8*(4u-to+i));
		}
#pragma unroll
		for (int v=0; v<SHAPE_MAXVARIANTS; ++v)
		{
			if ((u32)v < nVar)
			{
				const u32 var = uni( P.shapeVariants[ v]);
				const u32 kind = var & 3u, o = (var >> 2) & 3u, k = (var >> 4) & 7u;
				const u32 kmask = k >= 4u ? 0xFFFFFFFFu : ((1u << (8*k)) - 1u);
				if (emit)
				{
					u32 tag = 0, key = 0;
					if (kind == (u32)SHAPE_PREVWORD) { if (qLen >= 1u && qLen <= 64u) { tag = (u32)SHAPE_PREVWORD | (qLen << 8); key = literalHashFinish( qH); } }
					else if (kind == (u32)SHAPE_PREFIX)
					{
						if (pLen >= o + k)
						{
							tag = var;
							if (o + k <= 4u) key = (first4 >> (8*o)) & kmask;
							else for (u32 i=0; i<k; ++i) key |= (u32)w.doc[ pFrom + o + i] << (8*i);
						}
					}
					else if (pLen >= k) { tag = var; key = k >= 4u ? last4 : (last4 >> (8*(4u-k))); }
					if (tag)
					{
						const u32 fp = shapeFingerprint( tag, key, P.shapeSalt);
						u32 slot = shapeSlotHash( tag, key) & P.shapeMask;
						for (u32 probes=0; probes<=P.shapeMask; ++probes)
						{
							const u64 e = T.at( P.shapeFpOffset + slot);		// fingerprint | (count << 24 | pattern or list) << 32
							if (!(u32)e) break;
							if ((u32)e == fp) { lb[ 1+v] = (u32)(e >> 32) & 0xFFFFFFu; lc[ 1+v] = (u32)(e >> 56); break; }
							slot = (slot+1) & P.shapeMask;
						}
					}
				}
			}
		}
	}
#pragma unroll
	for (int k=0; k<NLIST; ++k) total += lc[ k];
	// ---- every lane reserves its records; 64 candidates a round get a lane each for the walk that confirms them
	const u32 incl = waveScanAdd( total);
	const u32 waveTotal = (u32)__builtin_amdgcn_readlane( incl, 63);
	if (!waveTotal) return;
	if (w.nQueue + waveTotal > w.queueCap) { w.err = L1D_ERR_ARENA; return; }
	uint4* stage = (uint4*)ends;
	// The usual lane: no list longer than one pattern, no more than four candidates.  Then its candidates are a handful of keys
	// (pattern << 1 | literal) in registers, ordered by a five-comparator network; the others merge their lists by pattern index.
	bool simple = total <= 4u;
#pragma unroll
	for (int k=0; k<NLIST; ++k) if (lc[ k] > 1u) simple = false;
	u32 c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu, c3 = 0xFFFFFFFFu;
	u32 head[ NLIST];
	if (simple)
	{
		u32 nc = 0;
#pragma unroll
		for (int k=0; k<NLIST; ++k)
		{
			if (lc[ k])
			{
				const u32 key = k == 0 ? ((lit0 << 1) | 1u) : (lb[ k] << 1);
				c0 = nc == 0 ? key : c0; c1 = nc == 1 ? key : c1; c2 = nc == 2 ? key : c2; c3 = nc == 3 ? key : c3;
				++nc;
			}
		}
#define SPA_CSWAP( A, B) { const u32 lo = A < B ? A : B, hi = A < B ? B : A; A = lo; B = hi; }
		SPA_CSWAP( c0, c1) SPA_CSWAP( c2, c3) SPA_CSWAP( c0, c2) SPA_CSWAP( c1, c3) SPA_CSWAP( c1, c2)
#undef SPA_CSWAP
	}
	const bool anyGeneral = __ballot( !simple && total) != 0;
	if (anyGeneral)
	{
#pragma unroll
		for (int k=0; k<NLIST; ++k)
		{
			head[ k] = 0xFFFFFFFFu;
			if (!simple && lc[ k]) head[ k] = k == 0 ? lit0 : (lc[ k] == 1u ? lb[ k] : P.shapePats[ lb[ k]]);	// (a shape entry of one pattern holds the pattern in place of the list)
		}
	}
	const u32 base = incl - total;
	u32 produced = 0;
	for (u32 r0=0; r0<waveTotal; r0+=64)
	{
		while (produced < total && base + produced < r0 + 64u)
		{
			u32 cand;
			if (simple)
			{
				const u32 key = produced == 0 ? c0 : (produced == 1 ? c1 : (produced == 2 ? c2 : c3));
				cand = (key >> 1) | ((key & 1u) ? (u32)L1_LITERAL_FLAG : 0u);
			}
			else
			{
				u32 best = 0xFFFFFFFFu; int which = 0;
#pragma unroll
				for (int k=0; k<NLIST; ++k) if (head[ k] < best) { best = head[ k]; which = k; }
#pragma unroll
				for (int k=0; k<NLIST; ++k)
				{
					if (k == which)
					{
						if (k != 0 && lc[ k] == 1u) { lc[ k] = 0; head[ k] = 0xFFFFFFFFu; }
						else
						{
							++lb[ k]; --lc[ k];
							head[ k] = lc[ k] ? (k == 0 ? P.litPats[ lb[ 0]] : P.shapePats[ lb[ k]]) : 0xFFFFFFFFu;
						}
					}
				}
				cand = which == 0 ? (best | (u32)L1_LITERAL_FLAG) : best;
			}
			stage[ base + produced - r0] = make_uint4( to, cand, pFrom, 0u);
			++produced;
		}
		__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
		const u32 nr = (waveTotal - r0) < 64u ? (waveTotal - r0) : 64u;
		if (LANE < nr)
		{
			const uint4 cnd = stage[ LANE];
			u32 fromOut = cnd.z, flags = (u32)L1_LITERAL_FLAG;
			const u32 pi = cnd.y & ~(u32)L1_LITERAL_FLAG;
			if (!(cnd.y & (u32)L1_LITERAL_FLAG))
			{
				const uint2 pw = *(const uint2*)((const u32*)&P.patterns[ pi] + 5);	// {maskLo, maskHi}
				const u32 word = ((const u32*)&P.patterns[ pi])[ 1];
				fromOut = confirmWalk<LDS>( w.doc, len, P, T, ring, ringLo, word, ((u64)pw.y << 32) | pw.x, cnd.x);
				if (fromOut == cnd.x) flags |= (u32)L1_DEAD_FLAG;
			}
			*(uint4*)(w.queue + 4*(u64)(w.nQueue + r0 + LANE)) = make_uint4( cnd.x, pi | flags, fromOut, 0u);
		}
		__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
	}
	w.nQueue += waveTotal;
}

// The runs of word characters of a unit, a lane per byte: where a run ends the lane knows the run's start, length and hash and those of
// the word before it.  Only one byte in six ends a run, so the ends are collected -- up to 64 of them, over several tiles -- before
// they are probed with a lane each (wordsFlush).
template <bool LDS>
__device__ __forceinline__ void wordsUnit( LexWave& w, const L1Params& P, const LexTab<LDS>& T, const u32 segBeg, const u32 segEnd, unsigned short* ring, u32* ends)
{
	const u32 len = w.docLen;
	u32 ctxReg = 0, clsReg = 0;		// byte -> context, class: lane l keeps the entries of bytes 4l..4l+3
	for (u32 k=0; k<4; ++k) { const u32 cl = P.byteClass[ 4*LANE + k]; clsReg |= cl << (8*k); ctxReg |= (u32)P.classCtx[ cl] << (8*k); }
	auto isWordAt = [&]( u32 pos) -> bool { return P.classCtx[ P.byteClass[ uni( (u32)w.doc[ pos])]] == (u32)CTX_WORD; };
	// where to begin: before the unit, at the start of a run (or between runs), so that every run that ends inside the unit is
	// seen from its first byte -- and one run further back when that one could be the word before a run of the unit
	u32 w0 = segBeg > 160u ? segBeg - 160u : 0u;
	if (w0 > 0 && isWordAt( w0-1)) w0 = runStartBefore( P, w.doc, w0, ctxReg);
	if (w0 >= 2 && isWordAt( w0-2)) w0 = runStartBefore( P, w.doc, w0-1, ctxReg);
	WordCarry c; c.in = false; c.hash = 0; c.len = 0; c.start = 0; c.lastValid = false; c.lastTo = 0; c.lastHash = 0; c.lastLen = 0;
	const u32 nVar = uni( P.nofShapeVariants);
	u32 nEnds = 0, firstTile = 0;		// ends that wait, the tile the oldest of them was seen in
	u32 ahead = (w0 + LANE < len) ? (u32)w.doc[ w0 + LANE] : 0u;
	u32 tile = w0;
	for (; tile<=len && tile<=segEnd && !w.err; tile+=64)
	{
		const u32 inTile = (len - tile) < 64 ? (len - tile) : 64;
		const u32 mine = ahead;			// (the next tile's bytes are requested one tile ahead)
		ahead = (tile + 64u + LANE < len) ? (u32)w.doc[ tile + 64u + LANE] : 0u;
		const u32 ctxL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((mine >> 2) << 2), (int)ctxReg) >> ((mine & 3u)*8)) & 0xFFu;
		const u32 clsL = ((u32)__builtin_amdgcn_ds_bpermute( (int)((mine >> 2) << 2), (int)clsReg) >> ((mine & 3u)*8)) & 0xFFu;
		ring[ (tile + LANE) & (WORD_RING-1)] = (unsigned short)(clsL | (ctxL << 8));		// (the ring keeps the last 512 bytes)
		const bool isW = LANE < inTile && ctxL == (u32)CTX_WORD;
		// ---- the runs of the tile (as tileLiterals)
		const u64 wm = __ballot( isW);
		const u64 prevW = (wm << 1) | (c.in ? 1ull : 0ull);
		const u64 startM = wm & ~prevW;
		const u64 endMask = ~wm & prevW;			// bit e: the run that ended just before byte tile+e
		const u64 sb = startM & (((1ull << LANE) - 1ull) | (1ull << LANE));
		const int ss = sb ? 63 - (int)__builtin_clzll( sb) : -1;
		u32 H = isW ? mine + 1u : 0u, M = (u32)L1_LITHASH_MUL;
#pragma unroll
		for (int d=1; d<64; d<<=1)
		{
			const u32 Hs = (u32)__shfl_up( (int)H, d), Ms = (u32)__shfl_up( (int)M, d);
			const bool take = isW && (int)LANE >= d && (int)LANE - d >= ss;
			if (take) { H = Hs * M + H; M = Ms * M; }
		}
		const bool carried = isW && ss < 0;
		const u32 Htot = carried ? c.hash * M + H : H;
		u32 runLen = isW ? (carried ? c.len + LANE + 1u : LANE - (u32)ss + 1u) : 0u;
		if (runLen > 65u) runLen = 65u;
		const u32 from = carried ? c.start : tile + (u32)(ss < 0 ? 0 : ss);
		u32 pH = (u32)__shfl_up( (int)Htot, 1), pLen = (u32)__shfl_up( (int)runLen, 1), pFrom = (u32)__shfl_up( (int)from, 1);
		if (LANE == 0) { pH = c.hash; pLen = c.len; pFrom = c.start; }
		const bool isEnd = (endMask >> LANE) & 1ull;
		const u32 to = tile + LANE;
		// ---- the run before mine, when it ended one byte before mine begins (PREVWORD)
		u32 qH = 0, qLen = 0;
		{
			const u32 want = pFrom - 1u;			// its end offset
			const u32 src = want - tile;			// lane that saw it end, if in this tile
			const u32 sH = (u32)__builtin_amdgcn_ds_bpermute( (int)((src & 63u) << 2), (int)pH);
			const u32 sLen = (u32)__builtin_amdgcn_ds_bpermute( (int)((src & 63u) << 2), (int)pLen);
			if (isEnd && pFrom >= 2u)
			{
				if (want >= tile) { if ((endMask >> (src & 63u)) & 1ull) { qH = sH; qLen = sLen; } }
				else if (c.lastValid && c.lastTo == want) { qH = c.lastHash; qLen = c.lastLen; }
			}
		}
		// ---- the ends that count go behind the ones that wait
		const bool emit = isEnd && pLen >= 1u && ((to >= segBeg && to < segEnd) || (to == segEnd && segEnd == len));
		const u64 em = __ballot( emit);
		if (em)
		{
			const u32 rank = (u32)__builtin_amdgcn_mbcnt_hi( (u32)(em >> 32), __builtin_amdgcn_mbcnt_lo( (u32)em, 0));
			if (emit)
			{
				u32* e = ends + WORD_ENDWORDS*(nEnds + rank);
				e[ 0] = to; e[ 1] = pFrom; e[ 2] = pH; e[ 3] = qH; e[ 4] = (pLen > 65u ? 65u : pLen) | ((qLen > 65u ? 65u : qLen) << 8);
			}
			if (!nEnds) firstTile = tile;
			nEnds += (u32)__builtin_popcountll( em);
		}
		__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
		// ---- carry: the run that reaches the end of the tile, the last run that ended
		c.in = (wm >> 63) & 1ull;
		if (c.in)
		{
			c.hash = uni( (u32)__shfl( (int)Htot, 63)); c.len = uni( (u32)__shfl( (int)runLen, 63)); c.start = uni( (u32)__shfl( (int)from, 63));
		}
		if (endMask)
		{
			const u32 e = 63u - (u32)__builtin_clzll( endMask);
			c.lastValid = true; c.lastTo = tile + e;
			c.lastHash = (u32)__builtin_amdgcn_readlane( pH, e); c.lastLen = (u32)__builtin_amdgcn_readlane( pLen, e);
		}
		// ---- a full batch, or ends about to lose the bytes before them from the ring: probe
		const bool pressure = nEnds && tile + 64u - firstTile > (u32)WORD_RING - 192u;
		while (!w.err && (nEnds >= 64u || (pressure && nEnds)))
		{
			const u32 count = nEnds < 64u ? nEnds : 64u;
			const u32 ringLo = tile + 64u > w0 + (u32)WORD_RING ? tile + 64u - (u32)WORD_RING : w0;
			// (what is behind the batch is read before the candidates of the batch are written over it... it lies behind them: entries 64..)
			wordsFlush<LDS>( w, P, T, ring, ringLo, ends, count, nVar);
			const u32 rest = nEnds - count;
			u32 keep[ WORD_ENDWORDS];
			if (LANE < rest) { _Pragma("unroll") for (int k=0; k<WORD_ENDWORDS; ++k) keep[ k] = ends[ WORD_ENDWORDS*(64u + LANE) + k]; }
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
			if (LANE < rest) { _Pragma("unroll") for (int k=0; k<WORD_ENDWORDS; ++k) ends[ WORD_ENDWORDS*LANE + k] = keep[ k]; }
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
			nEnds = rest; firstTile = tile;
		}
	}
	// the ends that still wait (the ring holds the bytes up to the last tile)
	if (!w.err && nEnds)
	{
		const u32 doneTo = tile;	// (one behind the last tile's first byte + 64... the loop has stepped past it)
		const u32 ringLo = doneTo > w0 + (u32)WORD_RING ? doneTo - (u32)WORD_RING : w0;
		wordsFlush<LDS>( w, P, T, ring, ringLo, ends, nEnds, nVar);
	}
}

template <bool LDS>
__device__ __forceinline__ void wordsDocuments( const L1Params& P, unsigned short* ring, u32* ends)
{
	if (!P.wordsKernel) return;
	LexTab<LDS> T;
	stageTables( P, T);
	const u32 chunked = ldu( (const u32*)&P.counters[ L1C_CHUNKED]);
	LexWave w;
	w.events = 0; w.nEvents = 0; w.cnt = 0; w.tailPos = 0; w.tailEnd = 0;
	const u32 nunits = ldu( (const u32*)&P.counters[ L1C_UNITS]);
	for (u32 round=0; round<=nunits; ++round)
	{
		u32 unit = 0;
		if (LANE == 0) unit = atomicAdd( (u32*)&P.counters[ L1C_CURSOR4], 1u);
		unit = uni( unit);
		if (unit >= nunits) break;
		u32 doc = unit;
		if (chunked)
		{
			u32 lo = 0, hi = P.ndocs;			// last document whose first unit is <= unit
			while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (ldu( &P.unitStart[ mid]) <= unit) lo = mid; else hi = mid; }
			doc = lo;
		}
		u64 beg, end;
		docBounds( P, doc, beg, end);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		u32 segBeg = 0, segEnd = w.docLen;
		if (chunked)
		{
			segBeg = (unit - ldu( &P.unitStart[ doc])) * P.chunkBytes;
			segEnd = (w.docLen - segBeg) < P.chunkBytes ? w.docLen : segBeg + P.chunkBytes;
		}
		const u64 qb = queueBase( P, beg + segBeg, unit);
		w.queue = P.wordQueue + 4*qb;
		w.queueCap = (u32)(queueBase( P, beg + segEnd, unit + 1) - qb);
		w.nQueue = 0; w.err = 0;
		wordsUnit<LDS>( w, P, T, segBeg, segEnd, ring, ends);
		if (LANE == 0)
		{
			P.wordCount[ unit] = w.err ? 0u : w.nQueue;
			if (w.err) P.docStatus[ doc] = (int32_t)w.err;
			if (w.err == L1D_ERR_ARENA) atomicAdd( (unsigned long long*)&P.counters[ L1C_OVER_QUEUE], 1ull);
			atomicAdd( (unsigned long long*)&P.counters[ L1C_WORDREPORTS], (unsigned long long)w.nQueue);
		}
	}
}

// ---------------------------------------------------------------- stages 2-3 with the words kernel: two queues, merged
// The automaton's reports (scan kernel) and the word reports (words kernel) of a document, both in (end offset, pattern) order,
// go through the reference's handler in the order the reference's callback sees them.  Each queue is read 64 records at a time
// (the automaton's get their leftmost starts on the way, nextBatch); a chunked document has one slice per chunk in either queue.
struct ReportStream
{
	const u32* queue; u32 nq, qi, qb, qn, unit;
	LaneReport lr;
};
template <bool LDS, bool CP, bool CH, bool WORDS>
__device__ __forceinline__ void streamSlice( ReportStream& s, const LexWave& w, const L1Params& P)
{
	const u64 qbase = ((w.docBegin + (u64)(s.unit - w.unit0) * P.chunkBytes) * P.queueMul >> 4) + 64ull*s.unit;
	s.queue = (WORDS ? P.wordQueue : P.reportQueue) + 4*qbase;
	s.nq = ldu( WORDS ? &P.wordCount[ s.unit] : &P.reportCount[ s.unit]);
	s.qi = 0; s.qb = 0; s.qn = 0;
}
template <bool LDS, bool CP, bool CH, bool WORDS>
__device__ __forceinline__ void streamFill( ReportStream& s, const LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	// the next batch of the current slice, or of the next slice that holds any
	while (s.qi == s.nq)
	{
		if (!CH || s.unit + 1u >= w.unitEnd) return;
		++s.unit;
		streamSlice<LDS,CP,CH,WORDS>( s, w, P);
	}
	s.qb = s.qi;
	if (WORDS)
	{
		s.qn = (s.nq - s.qb) < 64u ? (s.nq - s.qb) : 64u;
		resolveStarts<LDS,CP>( s.queue, w.doc, w.docLen, P, T, s.qb, s.qn, s.lr);
	}
	else nextBatch<LDS,CP>( s.queue, w.doc, w.docLen, P, T, s.nq, s.qb, s.qn, s.lr);
}
template <bool LDS, bool CP, bool CH>
__device__ void postDocumentMerged( LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	ReportStream A, B;
	A.unit = w.unit0; B.unit = w.unit0;
	A.lr.to = 0; A.lr.from = 0; A.lr.id = 0; A.lr.levelBind = 0; A.lr.prefixLen = 0; A.lr.suffixLen = 0; A.lr.pi = 0; A.lr.def = 0; A.lr.skip = 0;
	B.lr = A.lr;
	streamSlice<LDS,CP,CH,false>( A, w, P); streamFill<LDS,CP,CH,false>( A, w, P, T);
	streamSlice<LDS,CP,CH,true>( B, w, P); streamFill<LDS,CP,CH,true>( B, w, P, T);
	const u64 NOKEY = ~0ull;
	while (!w.err)
	{
		const u32 xa = A.qi - A.qb, xb = B.qi - B.qb;
		u64 keyA = NOKEY, keyB = NOKEY;
		if (A.qi < A.nq) keyA = ((u64)(u32)__builtin_amdgcn_readlane( A.lr.to, xa) << 32) | (u32)__builtin_amdgcn_readlane( A.lr.pi, xa);
		if (B.qi < B.nq) keyB = ((u64)(u32)__builtin_amdgcn_readlane( B.lr.to, xb) << 32) | (u32)__builtin_amdgcn_readlane( B.lr.pi, xb);
		if (keyA == NOKEY && keyB == NOKEY) break;
		if (keyB < keyA)
		{
			if (!__builtin_amdgcn_readlane( B.lr.skip, xb))
			{
				handleReport( w, P, (u32)__builtin_amdgcn_readlane( B.lr.id, xb), (u32)__builtin_amdgcn_readlane( B.lr.levelBind, xb),
					(u32)__builtin_amdgcn_readlane( B.lr.prefixLen, xb), (u32)__builtin_amdgcn_readlane( B.lr.suffixLen, xb),
					(u32)__builtin_amdgcn_readlane( B.lr.from, xb), (u32)(keyB >> 32));
			}
			++B.qi;
			if (B.qi == B.qb + B.qn || B.qi == B.nq) streamFill<LDS,CP,CH,true>( B, w, P, T);
		}
		else
		{
			if (!__builtin_amdgcn_readlane( A.lr.skip, xa))
			{
				handleReport( w, P, (u32)__builtin_amdgcn_readlane( A.lr.id, xa), (u32)__builtin_amdgcn_readlane( A.lr.levelBind, xa),
					(u32)__builtin_amdgcn_readlane( A.lr.prefixLen, xa), (u32)__builtin_amdgcn_readlane( A.lr.suffixLen, xa),
					(u32)__builtin_amdgcn_readlane( A.lr.from, xa), (u32)(keyA >> 32));
			}
			++A.qi;
			if (A.qi == A.qb + A.qn || A.qi == A.nq) streamFill<LDS,CP,CH,false>( A, w, P, T);
		}
	}
}

// ---------------------------------------------------------------- stage 3, a cluster of reports per lane (round 3)
// The reference's handler (patternLexer.cpp:727-822) looks at its event array from the back only: the delete pass stops at the first event
// that starts left of the new match, the ignore pass at the first that ends left of its end, the insertion at the first that does not
// start right of it.  Take the reports in callback order r0 r1 ... and say there is a BOUNDARY before ri when every report from ri on
// starts right of every earlier report's start and ends right of every earlier report's end: then none of the three scans of a report
// behind the boundary ever gets past the events the reports behind the boundary have left -- the handler works on the events of its own
// CLUSTER (the reports between two boundaries) as if the array began there, and what a cluster leaves is appended to the array as it is.
// A window of 64 merged reports has its boundaries found with two prefix maxima and two suffix minima; every complete cluster of the
// window is then run through the handler by ONE LANE on a little event array of its own in LDS, all clusters at once, and the survivors
// are appended to the wave's event array with one prefix sum.  What the lanes cannot take goes through handleReport one report after
// the other, as before: the reports in front of the window's first boundary (they may reach into the array), a cluster that leaves more
// than LOC_CAP events or holds a symbol lookup, a cluster as long as the window.  The last cluster of a window waits for the next one
// unless the input has ended (its last reports may still be to come).
#ifndef SPA_L1_POST_LOC_CAP
#define SPA_L1_POST_LOC_CAP 6
#endif
enum {WIN_CAP=192, LOC_CAP=SPA_L1_POST_LOC_CAP, REC_SYMBOL=1u<<16, REC_SIZEERR=1u<<31};

__device__ __forceinline__ u32 waveSuffixMin( u32 v)		// inclusive, identity ~0
{
	u32 t;
	t = (u32)__builtin_amdgcn_update_dpp( -1, (int)v, 0x101, 0xF, 0xF, false); v = t < v ? t : v;	// row_shl:1
	t = (u32)__builtin_amdgcn_update_dpp( -1, (int)v, 0x102, 0xF, 0xF, false); v = t < v ? t : v;
	t = (u32)__builtin_amdgcn_update_dpp( -1, (int)v, 0x104, 0xF, 0xF, false); v = t < v ? t : v;
	t = (u32)__builtin_amdgcn_update_dpp( -1, (int)v, 0x108, 0xF, 0xF, false); v = t < v ? t : v;
	const u32 m1 = (u32)__builtin_amdgcn_readlane( v, 16), m2 = (u32)__builtin_amdgcn_readlane( v, 32), m3 = (u32)__builtin_amdgcn_readlane( v, 48);
	const u32 s2 = m2 < m3 ? m2 : m3, s1 = m1 < s2 ? m1 : s2;
	const u32 behind = LANE < 16u ? s1 : (LANE < 32u ? s2 : (LANE < 48u ? m3 : ~0u));
	return behind < v ? behind : v;
}

// the record of a resolved report as the handler takes it: {end, start, lexem id, level|posbind<<8|flags}, sub expression selection
// applied (patternLexer.cpp:731-741); false: the report is dropped
__device__ __forceinline__ bool preparedRecord( const LaneReport& lr, uint4& rec)
{
	u32 from = lr.from, to = lr.to;
	u32 fl = lr.levelBind & 0x1FFFFu;
	if (to - from >= 65535u) fl |= (u32)REC_SIZEERR;						// :727-730, raised when the handler gets there
	bool live = !lr.skip;
	if (lr.levelBind & (1u<<17))
	{
		if (lr.prefixLen + lr.suffixLen > to - from) live = (fl & (u32)REC_SIZEERR) != 0;
		else { from += lr.prefixLen; to -= lr.suffixLen; }
	}
	rec = make_uint4( to, from, lr.id, fl);
	return live;
}

// ranks of the live records of two sorted batches in their merge (keys are distinct: a pattern reports through one of the queues only)
__device__ __forceinline__ void mergeRanks( u64 liveX, u32 xHi, u32 xLo, u32& rankX, bool isLiveY, u32 yHi, u32 yLo, u32& rankY)
{
	const u64 keyY = ((u64)yHi << 32) | yLo;
	while (liveX)
	{
		const u32 x = (u32)__builtin_ctzll( liveX);
		liveX &= liveX - 1;
		const u64 kx = ((u64)(u32)__builtin_amdgcn_readlane( xHi, x) << 32) | (u32)__builtin_amdgcn_readlane( xLo, x);
		const bool less = isLiveY && keyY < kx;
		const u32 c = (u32)__builtin_popcountll( __ballot( less));
		if (LANE == x) rankX += c;
		if (isLiveY && !less) ++rankY;
	}
}

template <bool LDS, bool CP, bool CH>
__device__ void postDocumentClusters( LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	__shared__ uint4 postWin[ 4][ WIN_CAP];
	__shared__ uint4 postLoc[ 4][ LOC_CAP*64];
	uint4* win = postWin[ (threadIdx.x >> 6) & 3u];
	uint4* loc = postLoc[ (threadIdx.x >> 6) & 3u] + LANE;		// event j of this lane's cluster: loc[ 64*j]
	ReportStream A, B;
	A.unit = w.unit0; B.unit = w.unit0;
	A.lr.to = 0; A.lr.from = 0; A.lr.id = 0; A.lr.levelBind = 0; A.lr.prefixLen = 0; A.lr.suffixLen = 0; A.lr.pi = 0; A.lr.def = 0; A.lr.skip = 0;
	B.lr = A.lr;
	streamSlice<LDS,CP,CH,false>( A, w, P); streamFill<LDS,CP,CH,false>( A, w, P, T);
	streamSlice<LDS,CP,CH,true>( B, w, P); streamFill<LDS,CP,CH,true>( B, w, P, T);
	u32 wi = 0, wn = 0;			// the window: win[ wi .. wn)
	u32 carryF = 0, carryT = 0;		// 1 + the greatest start / end of the reports handled so far (0: none)
	bool exhausted = false;
	const u64 NOKEY = ~0ull;
	auto sequential = [&]( const uint4& rec, u32 s, u32 e)
	{
		if (w.cnt == 0 && w.nEvents) { __builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront"); reloadLanes( w); readTail( w); }
		for (u32 x=s; x<e && !w.err; ++x)
		{
			const u32 fl = (u32)__builtin_amdgcn_readlane( rec.w, x);
			if (fl & (u32)REC_SIZEERR) { w.err = L1D_ERR_LEXEMSIZE; break; }
			handleReport( w, P, (u32)__builtin_amdgcn_readlane( rec.z, x), fl & 0x1FFFFu, 0, 0, (u32)__builtin_amdgcn_readlane( rec.y, x), (u32)__builtin_amdgcn_readlane( rec.x, x));
		}
	};
	while (!w.err)
	{
		// ---- refill: what is left of the window moves to its front, the next records of the two queues are merged in behind it
		while (wn - wi < 64u && !exhausted)
		{
			const bool haveA = A.qi < A.nq, haveB = B.qi < B.nq;
			if (!haveA && !haveB) { exhausted = true; break; }
			const u32 left = wn - wi;
			uint4 keep = make_uint4( 0, 0, 0, 0);
			if (LANE < left) keep = win[ wi + LANE];
			const u32 xa = A.qi - A.qb, xb = B.qi - B.qb;
			// nothing behind a batch's last key is known yet: records up to the smaller of the two last keys are taken
			u64 limit = NOKEY;
			if (haveA && (A.qb + A.qn < A.nq || (CH && A.unit + 1u < w.unitEnd)))
				limit = ((u64)(u32)__builtin_amdgcn_readlane( A.lr.to, A.qn-1) << 32) | (u32)__builtin_amdgcn_readlane( A.lr.pi, A.qn-1);
			if (haveB && (B.qb + B.qn < B.nq || (CH && B.unit + 1u < w.unitEnd)))
			{
				const u64 lb = ((u64)(u32)__builtin_amdgcn_readlane( B.lr.to, B.qn-1) << 32) | (u32)__builtin_amdgcn_readlane( B.lr.pi, B.qn-1);
				if (lb < limit) limit = lb;
			}
			const bool takeA = haveA && LANE >= xa && LANE < A.qn && ((((u64)A.lr.to << 32) | A.lr.pi) <= limit);
			const bool takeB = haveB && LANE >= xb && LANE < B.qn && ((((u64)B.lr.to << 32) | B.lr.pi) <= limit);
			uint4 recA, recB;
			const bool liveA = preparedRecord( A.lr, recA) && takeA, liveB = preparedRecord( B.lr, recB) && takeB;
			const u64 liveAm = __ballot( liveA), liveBm = __ballot( liveB);
			u32 rankA = (u32)__builtin_amdgcn_mbcnt_hi( (u32)(liveAm >> 32), __builtin_amdgcn_mbcnt_lo( (u32)liveAm, 0));
			u32 rankB = (u32)__builtin_amdgcn_mbcnt_hi( (u32)(liveBm >> 32), __builtin_amdgcn_mbcnt_lo( (u32)liveBm, 0));
			if (liveAm && liveBm)
			{
				if (__builtin_popcountll( liveAm) <= __builtin_popcountll( liveBm)) mergeRanks( liveAm, A.lr.to, A.lr.pi, rankA, liveB, B.lr.to, B.lr.pi, rankB);
				else mergeRanks( liveBm, B.lr.to, B.lr.pi, rankB, liveA, A.lr.to, A.lr.pi, rankA);
			}
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
			if (LANE < left) win[ LANE] = keep;
			if (liveA) win[ left + rankA] = recA;
			if (liveB) win[ left + rankB] = recB;
			__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
			wi = 0; wn = left + (u32)__builtin_popcountll( liveAm) + (u32)__builtin_popcountll( liveBm);
			const u32 tookA = (u32)__builtin_popcountll( __ballot( takeA)), tookB = (u32)__builtin_popcountll( __ballot( takeB));
			if (tookA) { A.qi += tookA; if (A.qi == A.qb + A.qn || A.qi == A.nq) streamFill<LDS,CP,CH,false>( A, w, P, T); }
			if (tookB) { B.qi += tookB; if (B.qi == B.qb + B.qn || B.qi == B.nq) streamFill<LDS,CP,CH,true>( B, w, P, T); }
			if (!tookA && !tookB) { w.err = L1D_ERR_INTERNAL; break; }
		}
		if (w.err) break;
		const u32 avail = (wn - wi) < 64u ? (wn - wi) : 64u;
		if (!avail) break;
		const bool valid = LANE < avail;
		uint4 rec = make_uint4( 0, 0, 0, 0);
		if (valid) rec = win[ wi + LANE];
		// ---- boundaries
		const u32 inclF = waveScanMax( valid ? rec.y + 1u : 0u), inclT = waveScanMax( valid ? rec.x + 1u : 0u);
		u32 exclF = laneFromBelow( inclF), exclT = laneFromBelow( inclT);
		exclF = exclF > carryF ? exclF : carryF; exclT = exclT > carryT ? exclT : carryT;
		const u32 sufF = waveSuffixMin( valid ? rec.y + 1u : ~0u), sufT = waveSuffixMin( valid ? rec.x + 1u : ~0u);
		const u64 bounds = __ballot( valid && sufF > exclF && sufT > exclT);
		const u32 first = bounds ? (u32)__builtin_ctzll( bounds) : 64u;
		auto carryTo = [&]( u32 upTo)		// the reports [0, upTo) of the view are done
		{
			if (!upTo) return;
			const u32 f = (u32)__builtin_amdgcn_readlane( inclF, upTo-1), t = (u32)__builtin_amdgcn_readlane( inclT, upTo-1);
			carryF = f > carryF ? f : carryF; carryT = t > carryT ? t : carryT;
		};
		if (first)
		{
			// the reports in front of the first boundary may reach into the array
			const u32 e = first < avail ? first : avail;
			sequential( rec, 0, e);
			carryTo( e); wi += e;
			continue;
		}
		const bool final = exhausted && avail == wn - wi;
		const u32 lastB = 63u - (u32)__builtin_clzll( bounds);
		const u32 procEnd = final ? avail : lastB;
		if (procEnd == 0)
		{
			// one cluster as long as the view
			sequential( rec, 0, avail);
			carryTo( avail); wi += avail;
			continue;
		}
		// ---- every complete cluster through the handler, one lane each
		const bool head = ((bounds >> LANE) & 1ull) != 0 && LANE < procEnd;
		u32 len = 0;
		{
			const u64 above = (bounds >> LANE) >> 1;
			const u32 nextB = above ? LANE + 1u + (u32)__builtin_ctzll( above) : 64u;
			len = (nextB < procEnd ? nextB : procEnd) - LANE;
		}
		u32 n = 0;
		bool hard = false;
		for (u32 k=0; __ballot( head && !hard && k < len) != 0; ++k)
		{
			if (head && !hard && k < len)
			{
				const uint4 r = win[ wi + LANE + k];
				if (r.w & ((u32)REC_SYMBOL | (u32)REC_SIZEERR)) hard = true;
				else
				{
					const u32 level = r.w & 0xFFu, from = r.y, lastPos = r.x;
					u32 nofDeletes = 0;
					// delete pass (:757-777)
					for (u32 q=n; q>0; --q)
					{
						const uint4 m = loc[ 64*(q-1)];
						if (!(m.y >= from)) break;
						const u32 mlevel = m.w & 0xFFu;
						if ((r.z == m.x && m.y == from && mlevel == level) || (mlevel < level && m.y + m.z <= lastPos))
						{
							for (u32 t=q-1; t+1<n; ++t) loc[ 64*t] = loc[ 64*(t+1)];
							--n; ++nofDeletes;
						}
					}
					bool ignored = false;
					if (!nofDeletes)
					{
						// ignore pass (:778-792)
						for (u32 q=n; q>0; --q)
						{
							const uint4 m = loc[ 64*(q-1)];
							if (!(m.y + m.z >= lastPos)) break;
							if ((m.w & 0xFFu) > level && m.y <= from) { ignored = true; break; }
						}
					}
					if (!ignored)
					{
						if (n == (u32)LOC_CAP) hard = true;
						else
						{
							// insert (:793-822)
							u32 at = n;
							for (; at>0; --at)
							{
								const uint4 m = loc[ 64*(at-1)];
								if (!(m.y > from)) break;
								loc[ 64*at] = m;
							}
							loc[ 64*at] = make_uint4( r.z, from, lastPos - from, r.w & 0xFFFFu);
							++n;
						}
					}
				}
			}
		}
		const u64 hardm = __ballot( head && hard);
		const u32 firstHard = hardm ? (u32)__builtin_ctzll( hardm) : 64u;
		// ---- what the clusters in front of the first one that has to go the slow way have left is appended
		const bool mine = head && LANE < firstHard;
		const u32 cnt = mine ? n : 0u;
		const u32 incl = waveScanAdd( cnt);
		const u32 total = (u32)__builtin_amdgcn_readlane( incl, 63);
		if (w.nEvents + total + 2u > P.eventCap)
		{
			if (LANE == 0) atomicAdd( (unsigned long long*)&P.counters[ L1C_OVER_EVENTS], 1ull);
			w.err = L1D_ERR_ARENA; break;
		}
		if (total)
		{
			if (w.cnt) spillLanes( w, 0);
			uint4* dst = (uint4*)w.events + w.nEvents + (incl - cnt);
			for (u32 j=0; j<(u32)LOC_CAP; ++j)
			{
				if (j < cnt) dst[ j] = loc[ 64*j];
			}
			w.nEvents += total;
		}
		if (firstHard < 64u)
		{
			const u32 hlen = (u32)__builtin_amdgcn_readlane( len, firstHard);
			if (total) __builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
			carryTo( firstHard);
			sequential( rec, firstHard, firstHard + hlen);
			carryTo( firstHard + hlen); wi += firstHard + hlen;
		}
		else { carryTo( procEnd); wi += procEnd; }
	}
	if (!w.err && (A.qi != A.nq || B.qi != B.nq)) w.err = L1D_ERR_INTERNAL;
}

// ---------------------------------------------------------------- stage 4: ordinal positions + output (:893-945)
// The reference walks the sorted event array once with a little state machine (patternLexer.cpp:893-945):
// up to the first content/unique event only successor-bound events are kept (position 1); from there on
// the ordinal position advances whenever a content/unique event starts to the right of all earlier
// ones, a unique event directly after a unique event is dropped, successor-bound events get position+1.
// Every piece of that state is a prefix quantity, so 64 events are handled per step: one coalesced
// load, a max-scan (start of the rightmost counted event so far), two sum-scans (ordinal position,
// output index), one coalesced store.  Pass A counts (the output is reserved once per document),
// pass B writes.
__device__ __forceinline__ u32 scanAdd( u32 v)
{
	return waveScanAdd( v);
}
__device__ __forceinline__ u32 scanMax( u32 v)
{
	return waveScanMax( v);
}
__device__ void emitLexems( LexWave& w, const L1Params& P, u32 doc)
{
	enum {BIND_CONTENT=0, BIND_SUCCESSOR=1, BIND_PREDECESSOR=2, BIND_UNIQUE=3};
	const u32 n = w.nEvents;
	// the first content/unique event
	u32 first = n;
	for (u32 base=0; base<n && first==n; base+=64)
	{
		const u32 i = base + LANE;
		u32 bind = 0xFFu;
		if (i < n) bind = (w.events[ i].levelBind >> 8) & 0xFFu;
		const u64 m = __ballot( bind == BIND_UNIQUE || bind == BIND_CONTENT);
		if (m) first = base + (u32)__builtin_ctzll( m);
	}
	u64 outBase = 0;
	u32 total = 0;
	if (first < n)
	{
		for (int phase=0; phase<2 && !w.err; ++phase)
		{
			u32 out = 0;			// lexems written so far
			u32 runMax = 0;			// start of the rightmost counted event so far (valid from `first` on)
			u32 runOrd = 1;			// ordinal position after the events so far
			u32 prevBind = BIND_CONTENT;	// bind of the event before this step's first one
			for (u32 base=0; base<n; base+=64)
			{
				const u32 i = base + LANE;
				uint4 e = make_uint4( 0, 0, 0, 0xFF00u);		// {id, origpos, origsize, levelBind}
				if (i < n) e = *(const uint4*)&w.events[ i];
				const u32 bind = (e.w >> 8) & 0xFFu;
				u32 pb = (u32)__shfl_up( (int)bind, 1);
				if (LANE == 0) pb = prevBind;
				const bool valid = i < n;
				const bool after = valid && i > first;			// the state machine proper
				const bool isFirst = valid && i == first;
				const bool dropped = after && bind == BIND_UNIQUE && pb == BIND_UNIQUE;
				const bool counted = (after && !dropped && (bind == BIND_UNIQUE || bind == BIND_CONTENT)) || isFirst;
				// start of the rightmost counted event strictly before me
				const u32 mine = counted ? e.y + 1u : 0u;		// +1: 0 = none
				const u32 inclMax = scanMax( mine);
				u32 exclMax = (u32)__shfl_up( (int)inclMax, 1);
				if (LANE == 0) exclMax = 0;
				if (runMax > exclMax) exclMax = runMax;
				const bool advances = counted && !isFirst && (e.y + 1u) > exclMax;
				const u32 inclOrd = scanAdd( advances ? 1u : 0u);
				const u32 ord = runOrd + inclOrd;			// ordinal position after me
				bool emit; u32 pos;
				if (!valid) { emit = false; pos = 0; }
				else if (i < first) { emit = (bind == BIND_SUCCESSOR); pos = 1; }
				else if (dropped) { emit = false; pos = 0; }
				else { emit = true; pos = (bind == BIND_SUCCESSOR) ? ord + 1u : ord; }
				const u32 inclOut = scanAdd( emit ? 1u : 0u);
				if (phase && emit)
				{
					u32* o = P.lexems + 4*(outBase + out + inclOut - 1u);
					*(uint4*)o = make_uint4( e.x, pos, e.y, e.z);
				}
				out += uni( (u32)__shfl( (int)inclOut, 63));
				runOrd += uni( (u32)__shfl( (int)inclOrd, 63));
				const u32 stepMax = uni( (u32)__shfl( (int)inclMax, 63));
				if (stepMax > runMax) runMax = stepMax;
				prevBind = uni( (u32)__shfl( (int)bind, 63));
			}
			if (!phase)
			{
				total = out;
				u64 b = 0;
				if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ L1C_LEXEMS], (unsigned long long)total);
				outBase = ((u64)uni( (u32)(b >> 32)) << 32) | uni( (u32)b);
				if (outBase + total > P.lexemCapacity) { w.err = L1D_ERR_OUTPUT; total = 0; }
				if (!total) break;
			}
		}
	}
	if (LANE == 0)
	{
		P.docRange[ 2*(u64)doc] = outBase; P.docRange[ 2*(u64)doc+1] = w.err ? 0 : total;
	}
}

// ---------------------------------------------------------------- the two kernels
__device__ __forceinline__ void docBounds( const L1Params& P, u32 doc, u64& beg, u64& end)
{
	beg = ((u64)ldu( (const u32*)&P.docOffsets[ doc]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc]);
	end = ((u64)ldu( (const u32*)&P.docOffsets[ doc+1]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc+1]);
}

// ---------------------------------------------------------------- approximate literal tables
// A table with an edit distance expression (`literal ~N`) takes the reference's other route (src/patternLexer.cpp
// :333-412): every expression is pre-matched on the one-byte-per-character hash of the text
// (src/unicodeUtils.cpp:19-44) -- approximately for `~N` -- and every candidate is re-matched on the characters
// themselves (:450-601).  One wave per document: the characters are decoded once into charCp/charPos; then 64
// candidate end positions at a time, each lane runs both stages for its end position and every pattern; the
// events of the tile go through the handler in the order of the candidates (end position, pattern index).
// The second stage's choice among several approximate matches is a model pinned by the reference's two vectors,
// stated with the oracle's restatement (oracle/l1_oracle.cpp, "approximate literal tables"); the code here
// follows that statement step by step.
__shared__ uint2 approxCand[ L1_APPROX_MAXPATTERNS][ 64];

__device__ __forceinline__ u32 oneByteHash( u32 cp) { return cp <= 127u ? cp : 128u + (cp & 127u); }
// lenient UTF-8: a lead byte with all its continuation bytes below `bound` is one character, any other byte is one of its own value
__device__ __forceinline__ void decodeAt( const unsigned char* doc, u32 bound, u32 at, u32& cp, u32& n)
{
	const u32 c = doc[ at];
	cp = c; n = 1;
	const u32 want = (c >= 0xC2u && c <= 0xDFu) ? 2u : (c >= 0xE0u && c <= 0xEFu) ? 3u : (c >= 0xF0u && c <= 0xF4u) ? 4u : 1u;
	if (want == 1u || at + want > bound) return;
	u32 v = c & (0xFFu >> (want+1u));
	for (u32 i=1; i<want; ++i)
	{
		const u32 x = doc[ at+i];
		if ((x & 0xC0u) != 0x80u) return;
		v = (v << 6) | (x & 0x3Fu);
	}
	cp = v; n = want;
}
__device__ __forceinline__ u32 encodeCp( u32 cp, u32* b)
{
	if (cp < 0x80u) { b[0] = cp; return 1; }
	if (cp < 0x800u) { b[0] = 0xC0u | (cp >> 6); b[1] = 0x80u | (cp & 0x3Fu); return 2; }
	if (cp < 0x10000u) { b[0] = 0xE0u | (cp >> 12); b[1] = 0x80u | ((cp >> 6) & 0x3Fu); b[2] = 0x80u | (cp & 0x3Fu); return 3; }
	b[0] = 0xF0u | (cp >> 18); b[1] = 0x80u | ((cp >> 12) & 0x3Fu); b[2] = 0x80u | ((cp >> 6) & 0x3Fu); b[3] = 0x80u | (cp & 0x3Fu); return 4;
}

// stage 1 for one end position: the smallest start s < e with levenshtein( hash(text[s:e]), hash(literal)) <= N
__device__ bool approxFirstStage( const u32* cps, const DevApproxPattern* ap, u32 e, u32& sOut)
{
	const u32 m = ap->len, N = ap->editdist;
	u32 L = m + N < e ? m + N : e;
	for (; L >= 1u && L + N >= m; --L)
	{
		u32 prev[ L1_APPROX_MAXCHARS+1], cur[ L1_APPROX_MAXCHARS+1];
		for (u32 j=0; j<=m; ++j) prev[ j] = j;
		for (u32 i=1; i<=L; ++i)
		{
			const u32 a = oneByteHash( cps[ e-L+i-1]);
			cur[ 0] = i;
			for (u32 j=1; j<=m; ++j)
			{
				u32 v = prev[ j-1] + (a != oneByteHash( ap->cp[ j-1]) ? 1u : 0u);
				if (prev[ j] + 1u < v) v = prev[ j] + 1u;
				if (cur[ j-1] + 1u < v) v = cur[ j-1] + 1u;
				cur[ j] = v;
			}
			for (u32 j=0; j<=m; ++j) prev[ j] = cur[ j];
		}
		if (prev[ m] <= N) { sOut = e - L; return true; }
	}
	return false;
}

// stage 2, `~N` expression: approximate search in the window [from, wend) of the document
__device__ bool approxSearch( const unsigned char* doc, u32 from, u32 wend, const DevApproxPattern* ap, u32& mFrom, u32& mTo)
{
	const u32 m = ap->len, maxCost = ap->editdist + 3u, NONE = 0xFFFFu;
	u32 cost[ L1_APPROX_MAXCHARS+1], start[ L1_APPROX_MAXCHARS+1], ncost[ L1_APPROX_MAXCHARS+1], nstart[ L1_APPROX_MAXCHARS+1];
	for (u32 k=0; k<=m; ++k) { cost[ k] = NONE; start[ k] = 0; }
	bool have = false; u32 best = NONE;
	u32 at = from;
	for (;;)
	{
		if (!have || best > 0u)
		{
			if (cost[ 0] == NONE || cost[ 0] > 0u) { cost[ 0] = 0; start[ 0] = at; }
		}
		for (u32 k=0; k<m; ++k)				// pattern characters skipped
		{
			if (cost[ k] == NONE) continue;
			const u32 c = cost[ k] + 1u;
			if (c > maxCost || (have && c >= best)) continue;
			if (cost[ k+1] == NONE || c < cost[ k+1]) { cost[ k+1] = c; start[ k+1] = start[ k]; }
		}
		if (at >= wend)
		{
			if (cost[ m] != NONE && (!have || cost[ m] < best)) { have = true; best = cost[ m]; mFrom = start[ m]; mTo = at; }
			break;
		}
		u32 ch, n;
		decodeAt( doc, wend, at, ch, n);
		at += n;
		for (u32 k=0; k<=m; ++k) ncost[ k] = NONE;
		for (u32 k=0; k<m; ++k)
		{
			if (cost[ k] == NONE) continue;
			{
				const u32 c = cost[ k] + (ch != ap->cp[ k] ? 1u : 0u);	// the character is the next pattern character, or stands for it
				if (c <= maxCost && !(have && c >= best))
				{
					if (ncost[ k+1] == NONE || c < ncost[ k+1]) { ncost[ k+1] = c; nstart[ k+1] = start[ k]; }
					if (k+1u == m && (!have || c < best)) { have = true; best = c; mFrom = start[ k]; mTo = at; }
				}
			}
			{
				const u32 c = cost[ k] + 1u;				// the character is an extra one
				if (c <= maxCost && !(have && c >= best))
				{
					if (ncost[ k] == NONE || c < ncost[ k]) { ncost[ k] = c; nstart[ k] = start[ k]; }
				}
			}
		}
		for (u32 k=0; k<=m; ++k) { cost[ k] = ncost[ k]; start[ k] = nstart[ k]; }
	}
	return have;
}

// stage 2, exact expression: the leftmost occurrence of the literal from the candidate's start on must end inside the candidate
__device__ bool exactSearch( const unsigned char* doc, u32 from, u32 to, const DevApproxPattern* ap, u32& mFrom, u32& mTo)
{
	u32 bytes[ 4*L1_APPROX_MAXCHARS];
	u32 nb = 0;
	for (u32 k=0; k<ap->len; ++k) nb += encodeCp( ap->cp[ k], bytes + nb);
	for (u32 at=from; at + nb <= to; ++at)
	{
		bool same = true;
		for (u32 q=0; q<nb && same; ++q) same = doc[ at+q] == bytes[ q];
		if (same) { mFrom = at; mTo = at + nb; return true; }
	}
	return false;
}

__device__ void approxDocument( LexWave& w, const L1Params& P, u32* cps, u32* cpos)
{
	const u32 len = w.docLen;
	// the characters of the document
	u32 nChars = 0;
	u64 carry = 0;				// continuation bytes at the start of the next tile that belong to a character of this one
	for (u32 tile=0; tile<len; tile+=64)
	{
		const u32 i = tile + LANE;
		u32 cp = 0, n = 0;
		if (i < len) decodeAt( w.doc, len, i, cp, n);
		const u64 m2 = __ballot( n >= 2u), m3 = __ballot( n >= 3u), m4 = __ballot( n >= 4u);
		const u64 covered = (m2 << 1) | (m3 << 2) | (m4 << 3) | carry;
		carry = (m2 >> 63) | (m3 >> 62) | (m4 >> 61);
		const u64 startM = __ballot( i < len) & ~covered;
		if ((startM >> LANE) & 1ull)
		{
			const u32 idx = nChars + (u32)__builtin_popcountll( startM & ((1ull << LANE) - 1ull));
			cps[ idx] = cp; cpos[ idx] = i;
		}
		nChars += (u32)__builtin_popcountll( startM);
	}
	if (LANE == 0) cpos[ nChars] = len;
	__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront");
	// candidates and their second stage, 64 end positions at a time
	for (u32 ct=0; ct<nChars && !w.err; ct+=64)
	{
		const u32 e = ct + 1u + LANE;
		for (u32 p=0; p<P.nofApprox; ++p)
		{
			const DevApproxPattern* ap = &P.approx[ p];
			uint2 ev = make_uint2( 0, 0);
			if (e <= nChars)
			{
				u32 s = 0;
				if (approxFirstStage( cps, ap, e, s))
				{
					const u32 from = cpos[ s], to = cpos[ e];
					u32 mf = 0, mt = 0;
					bool ok;
					if (ap->editdist)
					{
						u32 wend = to + 4u * ap->editdist;		// (:561: the candidate's bytes and editdist * sizeof(wchar_t) more)
						if (wend > len) wend = len;
						ok = approxSearch( w.doc, from, wend, ap, mf, mt);
					}
					else ok = exactSearch( w.doc, from, to, ap, mf, mt);
					if (ok && mt > mf) ev = make_uint2( mf, mt);
				}
			}
			approxCand[ p][ LANE] = ev;
		}
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront");
		const u32 inTile = (nChars - ct) < 64u ? (nChars - ct) : 64u;
		for (u32 l=0; l<inTile && !w.err; ++l)
		{
			for (u32 p=0; p<P.nofApprox && !w.err; ++p)
			{
				const uint2 ev = approxCand[ p][ l];
				const u32 mf = uni( ev.x), mt = uni( ev.y);
				if (!mt) continue;
				const DevApproxPattern* ap = &P.approx[ p];
				handleReport( w, P, ldu( &ap->id), ldu( &ap->levelBind), 0, 0, mf, mt);
			}
		}
		__builtin_amdgcn_fence( __ATOMIC_ACQ_REL, "wavefront");
	}
}

__device__ void approxDocuments( const L1Params& P)
{
	const u32 waveSlot = blockIdx.x;		// one wave per workgroup
	LexWave w;
	w.events = (Event*)(P.arenaBase + (u64)waveSlot * P.arenaWords);
	w.queue = 0; w.nQueue = 0; w.queueCap = 0;
#ifdef SPA_PROF
	for (int k=0; k<4; ++k) w.prof[ k] = 0;
#endif
	for (u32 round=0; round<=P.ndocs; ++round)
	{
		// every document comes from the device-side cursor (a workgroup that only becomes resident when others have
		// finished finds it exhausted and leaves at once: no document waits for a particular wave)
		u32 doc = 0;
		if (LANE == 0) doc = atomicAdd( (u32*)&P.counters[ L1C_CURSOR], 1u);
		doc = uni( doc);
		if (doc >= P.ndocs) break;
		u64 beg, end;
		docBounds( P, doc, beg, end);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		w.nEvents = 0; w.err = 0; w.cnt = 0; w.tailPos = 0; w.tailEnd = 0;
		w.e.id = 0; w.e.pos = 0; w.e.size = 0; w.e.lb = 0;
		approxDocument( w, P, P.charCp + beg + doc, P.charPos + beg + doc);
		spillLanes( w, 0);
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		if (!w.err) emitLexems( w, P, doc);
		else if (LANE == 0) { P.docRange[ 2*(u64)doc] = 0; P.docRange[ 2*(u64)doc+1] = 0; }
		if (LANE == 0)
		{
			P.docStatus[ doc] = (int32_t)w.err;
			atomicAdd( (unsigned long long*)&P.counters[ L1C_BYTES], (unsigned long long)w.docLen);
			if (w.err) atomicAdd( (unsigned long long*)&P.counters[ L1C_FAILED], 1ull);
		}
	}
}

template <bool LDS>
__device__ __forceinline__ void stageTables( const L1Params& P, LexTab<LDS>& T)
{
	T.g = P.tableImage; T.oChar = P.ldsChar; T.oAccept = P.ldsAccept; T.oStart = P.ldsStart; T.oShift = P.ldsShift; T.oSelf = P.ldsSelf;
	T.oExSrc = P.ldsExSrc; T.oExDst = P.ldsExDst;
	if (LDS)
	{
		// the workgroup stages the hot tables once
		for (u32 k=threadIdx.x; k<P.ldsWords; k+=blockDim.x) ldsImage[ k] = P.tableImage[ k];
		__syncthreads();
	}
}

// SCAN: automaton over the document's bytes, raw reports into the document's slice of the report queue
// UNITS: one wave counts the chunks of every document: unitStart[], the number of units, whether any document has more than
// one; and clears the per-document flags
__device__ void countUnits( const L1Params& P)
{
	u32 running = 0, chunked = 0;
	for (u32 base=0; base<P.ndocs; base+=64)
	{
		const u32 d = base + LANE;
		u32 n = 0;
		if (d < P.ndocs)
		{
			const u64 len = P.docOffsets[ d+1] - P.docOffsets[ d];
			n = len ? (u32)((len + P.chunkBytes - 1) / P.chunkBytes) : 1u;
			P.docStatus[ d] = 0; P.docSequential[ d] = 0;
		}
		const u32 incl = waveScanAdd( n);
		if (d < P.ndocs) P.unitStart[ d] = running + incl - n;
		running += uni( (u32)__shfl( (int)incl, 63));
		if (__ballot( n > 1u)) chunked = 1;
	}
	if (LANE == 0)
	{
		P.unitStart[ P.ndocs] = running;
		*(u32*)&P.counters[ L1C_UNITS] = running;
		*(u32*)&P.counters[ L1C_CHUNKED] = chunked;
	}
}


// SCAN: automaton over the bytes of a unit (a document, or a chunk of a long one), raw reports into the unit's slice of the
// report queue.  All sets of instances are launched, the ones that are not meant for the batch leave at once.
template <int PASSES, bool LDS, bool CP, bool CH>
__device__ void scanDocuments( const L1Params& P)
{
	// three sets of instances: plain (no classes by code point, no chunked document in the batch), chunks, classes by code point (+ chunks)
	const u32 chunked = ldu( (const u32*)&P.counters[ L1C_CHUNKED]);
	// (the sequential pass is launched on the _ch instance only and runs whatever the batch looks like: the lane-per-stream
	//  kernel cuts every unit into pieces, chunked batch or not)
	if (CP != (P.cpBlocks != 0 || P.nofNullable != 0) || (!CP && !P.sequentialPass && CH != (chunked != 0))) return;
	LexTab<LDS> T;
	stageTables( P, T);
	LexWave w;
	w.events = 0; w.nEvents = 0; w.cnt = 0; w.tailPos = 0; w.tailEnd = 0;
	const u32 nunits = P.sequentialPass ? P.ndocs : ldu( (const u32*)&P.counters[ L1C_UNITS]);
	// the loop is bounded so that it ends whatever the cursor holds
	for (u32 round=0; round<=nunits; ++round)
	{
		// every unit comes from the device-side cursor (a workgroup that only becomes resident when others have
		// finished finds it exhausted and leaves at once: no document waits for a particular wave)
		u32 unit = 0;
		if (LANE == 0) unit = atomicAdd( (u32*)&P.counters[ P.sequentialPass ? L1C_CURSOR3 : L1C_CURSOR], 1u);
		unit = uni( unit);
		if (unit >= nunits) break;
		u32 doc = unit, u0 = unit, u1 = unit + 1;
		if (CH && P.sequentialPass)
		{
			// the re-scan of the documents whose chunks could not be joined: in one piece, into the slices of all its units
			if (ldu( &P.docSequential[ doc]) == 0) continue;
			if (LANE == 0) atomicAdd( (unsigned long long*)&P.counters[ L1C_SEQDOCS], 1ull);
			u0 = ldu( &P.unitStart[ doc]); u1 = ldu( &P.unitStart[ doc+1]);
		}
		else if (CH && chunked)
		{
			u32 lo = 0, hi = P.ndocs;			// last document whose first unit is <= unit
			while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (ldu( &P.unitStart[ mid]) <= unit) lo = mid; else hi = mid; }
			doc = lo;
		}
		u64 beg, end;
		docBounds( P, doc, beg, end);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		u32 segBeg = 0, segEnd = w.docLen;
		if (CH && chunked && !P.sequentialPass)
		{
			segBeg = (unit - ldu( &P.unitStart[ doc])) * P.chunkBytes;
			segEnd = (w.docLen - segBeg) < P.chunkBytes ? w.docLen : segBeg + P.chunkBytes;
		}
		const u64 qb = queueBase( P, beg + segBeg, u0);
		w.queue = P.reportQueue + 4*qb;
		w.queueCap = (u32)(queueBase( P, beg + segEnd, u1) - qb);
		w.nQueue = 0; w.err = 0;
		scanDocument<PASSES,LDS,CP,CH>( w, P, T, segBeg, segEnd);
		if (LANE == 0)
		{
			if (w.err == L1D_CHUNK_UNPROVEN) { P.docSequential[ doc] = 1; P.reportCount[ u0] = 0; }
			else
			{
				P.reportCount[ u0] = w.err ? 0u : w.nQueue;
				for (u32 u=u0+1; u<u1; ++u) P.reportCount[ u] = 0;
				// (the sequential pass decides alone about its document: a chunk of the first pass may have left an error behind)
				if (w.err || P.sequentialPass) P.docStatus[ doc] = (int32_t)w.err;
				if (w.err == L1D_ERR_ARENA) atomicAdd( (unsigned long long*)&P.counters[ L1C_OVER_QUEUE], 1ull);
				atomicAdd( (unsigned long long*)&P.counters[ L1C_RAW], (unsigned long long)w.nQueue);
			}
		}
	}
}

// POST: literals, start of match, handler, ordinal positions, lexems
template <bool LDS, bool CP, bool CH>
__device__ void postDocuments( const L1Params& P)
{
	// (the same three sets of instances as the scan kernel's)
	const u32 chunked = ldu( (const u32*)&P.counters[ L1C_CHUNKED]);
	if (CP != (P.cpBlocks != 0 || P.nofNullable != 0) || (!CP && CH != (chunked != 0))) return;
	LexTab<LDS> T;
	stageTables( P, T);
	const u32 waveSlot = uni( blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
	LexWave w;
	w.events = (Event*)(P.arenaBase + (u64)waveSlot * P.arenaWords);
#ifdef SPA_PROF
	for (int k=0; k<4; ++k) w.prof[ k] = 0;
#endif
	for (u32 round=0; round<=P.ndocs; ++round)
	{
		// every document comes from the device-side cursor (a workgroup that only becomes resident when others have
		// finished finds it exhausted and leaves at once: no document waits for a particular wave)
		u32 doc = 0;
		if (LANE == 0) doc = atomicAdd( (u32*)&P.counters[ L1C_CURSOR2], 1u);
		doc = uni( doc);
		if (doc >= P.ndocs) break;
		u64 beg, end;
		docBounds( P, doc, beg, end);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		if (ldu( (const u32*)&P.docStatus[ doc]) != 0)
		{
			// failed in the scan kernel
			if (LANE == 0)
			{
				P.docRange[ 2*(u64)doc] = 0; P.docRange[ 2*(u64)doc+1] = 0;
				atomicAdd( (unsigned long long*)&P.counters[ L1C_BYTES], (unsigned long long)w.docLen);
				atomicAdd( (unsigned long long*)&P.counters[ L1C_FAILED], 1ull);
			}
			continue;
		}
		w.docBegin = beg;
		w.unit0 = chunked ? ldu( &P.unitStart[ doc]) : doc;
		w.unitEnd = chunked ? ldu( &P.unitStart[ doc+1]) : doc + 1u;
		w.unit = w.unit0;
		w.nEvents = 0; w.err = 0; w.cnt = 0; w.tailPos = 0; w.tailEnd = 0;
		w.e.id = 0; w.e.pos = 0; w.e.size = 0; w.e.lb = 0;
		const u64 tDoc = PROF_T();
		if (!CP && P.wordsKernel && P.postClusters) postDocumentClusters<LDS,CP,CH>( w, P, T);
		else if (!CP && P.wordsKernel) postDocumentMerged<LDS,CP,CH>( w, P, T);
		else
		{
			sliceOf( w, P);
			if (CH && !w.nQueue) (void)nextSlice( w, P);
			w.queueCap = w.nQueue;
			postDocument<LDS,CP,CH>( w, P, T);
		}
		spillLanes( w, 0);
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		if (!w.err) emitLexems( w, P, doc);
		else if (LANE == 0) { P.docRange[ 2*(u64)doc] = 0; P.docRange[ 2*(u64)doc+1] = 0; }
		PROF_ACC( 0, tDoc);
		if (LANE == 0)
		{
			P.docStatus[ doc] = (int32_t)w.err;
			atomicAdd( (unsigned long long*)&P.counters[ L1C_BYTES], (unsigned long long)w.docLen);
			if (w.err) atomicAdd( (unsigned long long*)&P.counters[ L1C_FAILED], 1ull);
		}
	}
#ifdef SPA_PROF
	if (LANE == 0) for (int k=0; k<4; ++k) atomicAdd( (unsigned long long*)&P.counters[ 4+k], (unsigned long long)w.prof[ k]);
#endif
}

} // anonymous namespace

// One scan instance per pass count (the per-pass rows of a byte step live in registers, so the count is a
// template parameter).
// (a second set, _cp, for batches with classes by code point -- \\p{..} sets, UCP -- or with documents scanned in chunks: the
// plain set pays nothing for either)
#define SPA_L1_KERNEL( NAME, N, T) \
extern "C" __global__ __launch_bounds__(T) void spa_l1_scan_kernel_##NAME( L1Params P) { if (P.ldsWords) scanDocuments<N,true,false,false>( P); else scanDocuments<N,false,false,false>( P); } \
extern "C" __global__ __launch_bounds__(T) void spa_l1_scan_kernel_##NAME##_ch( L1Params P) { if (P.ldsWords) scanDocuments<N,true,false,true>( P); else scanDocuments<N,false,false,true>( P); } \
extern "C" __global__ __launch_bounds__(T) void spa_l1_scan_kernel_##NAME##_cp( L1Params P) { if (P.ldsWords) scanDocuments<N,true,true,true>( P); else scanDocuments<N,false,true,true>( P); }
SPA_L1_KERNEL( p1, 1, 1024)
SPA_L1_KERNEL( p2, 2, 1024)
SPA_L1_KERNEL( p3, 3, 1024)
SPA_L1_KERNEL( p4, 4, 1024)
SPA_L1_KERNEL( p5, 5, 1024)
SPA_L1_KERNEL( p6, 6, 1024)
SPA_L1_KERNEL( p7, 7, 1024)
SPA_L1_KERNEL( p8, 8, 1024)
SPA_L1_KERNEL( p16, 16, 256)
SPA_L1_KERNEL( p32, 32, 256)
extern "C" __global__ __launch_bounds__(64) void spa_l1_approx_kernel( L1Params P) { approxDocuments( P); }
extern "C" __global__ __launch_bounds__(64) void spa_l1_units_kernel( L1Params P) { countUnits( P); }
extern "C" __global__ __launch_bounds__(256) void spa_l1_scan_lanes_kernel( L1Params P)
{
	if (P.scanWords <= 1) scanDocumentsLanes<1>( P); else if (P.scanWords == 2) scanDocumentsLanes<2>( P); else scanDocumentsLanes<4>( P);
}
// (12 waves: the table image of a large expression set leaves room for no more; 16 waves -- what 114 registers allow -- for small sets)
#define SPA_L1_WORDS_KERNEL( NAME, WAVES) \
extern "C" __global__ __launch_bounds__(64*WAVES) void NAME( L1Params P) \
{ \
	__shared__ unsigned short wordRing[ WAVES][ WORD_RING]; \
	__shared__ __attribute__((aligned(16))) u32 wordEnds[ WAVES][ WORD_ENDS*WORD_ENDWORDS]; \
	const u32 wv = threadIdx.x >> 6; \
	if (P.ldsWords) wordsDocuments<true>( P, wordRing[ wv], wordEnds[ wv]); else wordsDocuments<false>( P, wordRing[ wv], wordEnds[ wv]); \
}
SPA_L1_WORDS_KERNEL( spa_l1_words_kernel, L1_WORD_WAVES)
SPA_L1_WORDS_KERNEL( spa_l1_words_kernel_w16, L1_WORD_WAVES_SMALL)

enum {POST_WAVES=4};
// The post-processing kernel is bound by the latency of its dependent chains, not by issue slots: it gains from
// more resident waves as long as the register budget does not spill much.  Measured on 12288 x 64 KiB documents
// (tests/micro/sweep_l1_post.py): 100 registers / 16 waves per CU 72 ms, 96 / 20: 65 ms, 80 / 24: 59.4 ms,
// 72 / 28: 58.9 ms, 64 / 32: 100 ms (60 spills).
#ifndef SPA_L1_POST_WAVES_PER_EU
#define SPA_L1_POST_WAVES_PER_EU 6
#endif
#define SPA_L1_POST_OCC __attribute__((amdgpu_waves_per_eu( SPA_L1_POST_WAVES_PER_EU, SPA_L1_POST_WAVES_PER_EU)))
// (the two instances with the cluster-per-lane handler hold 9 KB of LDS per wave: 16 waves per CU)
#ifndef SPA_L1_POST_CLUSTER_WAVES_PER_EU
#define SPA_L1_POST_CLUSTER_WAVES_PER_EU 4
#endif
#define SPA_L1_POST_OCC4 __attribute__((amdgpu_waves_per_eu( SPA_L1_POST_CLUSTER_WAVES_PER_EU, SPA_L1_POST_CLUSTER_WAVES_PER_EU)))
extern "C" __global__ __launch_bounds__(64*POST_WAVES) SPA_L1_POST_OCC4 void spa_l1_post_kernel( L1Params P) { postDocuments<false,false,false>( P); }
extern "C" __global__ __launch_bounds__(64*POST_WAVES) SPA_L1_POST_OCC4 void spa_l1_post_kernel_ch( L1Params P) { postDocuments<false,false,true>( P); }
extern "C" __global__ __launch_bounds__(64*POST_WAVES) SPA_L1_POST_OCC void spa_l1_post_kernel_cp( L1Params P) { postDocuments<false,true,true>( P); }

namespace spa {
// PS: the parameters of the scan kernel (its table image holds the scanned passes only; nofPasses = 0: nothing to scan);
// PW: of the words kernel (the passes it walks + the shape table, in LDS when they fit; offsets biased); P: of the other kernels (all passes, read from global memory)
bool l1ScanByLanes( const L1Params& PS, const L1Params& P)
{
	return PS.nofPasses == 1 && PS.scanWords >= 1 && PS.scanWords <= 4 && PS.reportsOrdered && !P.cpBlocks && !P.nofNullable && PS.ldsWords && (size_t)PS.ldsWords * 8 <= 65536;
}

hipError_t launchL1Lex( const L1Params& PS, const L1Params& PW, const L1Params& P, unsigned nblocks, unsigned nthreads, unsigned laneBlocks, unsigned wordBlocks, unsigned wordWaves, unsigned postWaves, hipStream_t stream, hipEvent_t betweenKernels, hipEvent_t afterWords)
{
	if (P.nofApprox)
	{
		// approximate literal table: one kernel, one wave per workgroup
		hipLaunchKernelGGL( spa_l1_approx_kernel, dim3( postWaves), dim3( 64), 0, stream, P);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		if (betweenKernels) { e = hipEventRecord( betweenKernels, stream); if (e != hipSuccess) return e; }
		return afterWords ? hipEventRecord( afterWords, stream) : hipSuccess;
	}
	const size_t lds = (size_t)PS.ldsWords * 8;
	// the units of the batch (chunks of long documents), then BOTH sets of instances: which one a batch needs is known on
	// the device only (classes by code point: here; chunked documents: after the units kernel) -- the other one leaves at once
	hipLaunchKernelGGL( spa_l1_units_kernel, dim3( 1), dim3( 64), 0, stream, P);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	L1Params S = PS;
	S.sequentialPass = 1;
#define SPA_L1_LAUNCH_ONE( KERNEL, ARGS) do { \
	if (lds > 65536) { e = hipFuncSetAttribute( (const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; } \
	hipLaunchKernelGGL( KERNEL, dim3( nblocks), dim3( nthreads), lds, stream, ARGS); } while (0)
#define SPA_L1_LAUNCH( N) do { \
	if (P.cpBlocks || P.nofNullable) { SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_##N##_cp, PS); SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_##N##_cp, S); } \
	else { SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_##N, PS); SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_##N##_ch, PS); SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_##N##_ch, S); } } while (0)
	// what is left to scan fits four automaton words: a lane per stream (scanUnitLanes); the documents a piece of which could not
	// be joined go through the sequential pass of the one-pass instance behind it
	const bool lanes = l1ScanByLanes( PS, P);
	if (lanes)
	{
		hipLaunchKernelGGL( spa_l1_scan_lanes_kernel, dim3( laneBlocks), dim3( 256), lds, stream, PS);
		SPA_L1_LAUNCH_ONE( spa_l1_scan_kernel_p1_ch, S);
	}
	else switch (PS.nofPasses)
	{
		case 0: break;		// (nothing to scan: the caller has cleared the report counts)
		case 1: SPA_L1_LAUNCH( p1); break;
		case 2: SPA_L1_LAUNCH( p2); break;
		case 3: SPA_L1_LAUNCH( p3); break;
		case 4: SPA_L1_LAUNCH( p4); break;
		case 5: SPA_L1_LAUNCH( p5); break;
		case 6: SPA_L1_LAUNCH( p6); break;
		case 7: SPA_L1_LAUNCH( p7); break;
		case 8: SPA_L1_LAUNCH( p8); break;
		default:
			if (PS.nofPasses <= 16) SPA_L1_LAUNCH( p16);
			else if (PS.nofPasses <= 32) SPA_L1_LAUNCH( p32);
			else return hipErrorInvalidValue;
	}
	if (e != hipSuccess) return e;
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	if (betweenKernels) { e = hipEventRecord( betweenKernels, stream); if (e != hipSuccess) return e; }
	if (P.wordsKernel)
	{
		// (its own copy of the parameters: its image staged in LDS when it fits, PW.ldsWords; 16 or 12 waves per workgroup by what the image leaves)
		const size_t wlds = (size_t)PW.ldsWords * 8;
		if (wordWaves == (unsigned)L1_WORD_WAVES_SMALL)
		{
			if (wlds > 65536) { e = hipFuncSetAttribute( (const void*)spa_l1_words_kernel_w16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds); if (e != hipSuccess) return e; }
			hipLaunchKernelGGL( spa_l1_words_kernel_w16, dim3( wordBlocks), dim3( 64*L1_WORD_WAVES_SMALL), wlds, stream, PW);
		}
		else
		{
			if (wlds > 65536) { e = hipFuncSetAttribute( (const void*)spa_l1_words_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds); if (e != hipSuccess) return e; }
			hipLaunchKernelGGL( spa_l1_words_kernel, dim3( wordBlocks), dim3( 64*L1_WORD_WAVES), wlds, stream, PW);
		}
		e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	if (afterWords) { e = hipEventRecord( afterWords, stream); if (e != hipSuccess) return e; }
	// its own number of waves (one event array each), in workgroups of POST_WAVES
	if (P.cpBlocks || P.nofNullable) hipLaunchKernelGGL( spa_l1_post_kernel_cp, dim3( (postWaves + POST_WAVES-1) / POST_WAVES), dim3( 64*POST_WAVES), 0, stream, P);
	else
	{
		hipLaunchKernelGGL( spa_l1_post_kernel, dim3( (postWaves + POST_WAVES-1) / POST_WAVES), dim3( 64*POST_WAVES), 0, stream, P);
		hipLaunchKernelGGL( spa_l1_post_kernel_ch, dim3( (postWaves + POST_WAVES-1) / POST_WAVES), dim3( 64*POST_WAVES), 0, stream, P);
	}
	return hipGetLastError();
}
}
