// Level-1 pattern lexer on gfx950: one wavefront per document, one 64-bit automaton word per lane.
//
// What it replaces (reference, CPU): hs_scan (Intel Hyperscan, src/patternLexer.cpp:875-879), the
// per-match callback PatternLexerContext::match_event_handler (:717-826) and the ordinal-position
// pass of PatternLexerContext::match (:893-945).
//
// Stages per document, fused in one kernel so that raw matches never leave the wave's arena:
//  1. SCAN   bit-parallel position automaton (tables: l1_tables.h).  Lane l owns word l of every
//            pass; the document byte is wave-uniform, the table row of its byte class is one
//            coalesced 512-byte read; a report is detected with one __ballot per pass and byte.
//            Reports are appended to a queue in (end offset, pattern index) order = the order in
//            which the reference's handler is called.
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

enum {L1D_OK=0, L1D_ERR_ARENA=2, L1D_ERR_LEXEMSIZE=7, L1D_ERR_INTERNAL=8, L1D_ERR_OUTPUT=9};

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
	u32 oAccept, oStart, oShift, oSelf, oExSrc, oExDst;
	__device__ __forceinline__ u64 at( u32 off) const { if (LDS) return ldsImage[ off]; else return g[ off]; }
};

struct LexWave
{
	u32* queue;		// raw reports: 4 words {to, pattern, accLo, accHi}; after SOM word 2 holds `from`
	Event* events;
	u32 nQueue, nEvents, err;
	Event tail; bool tailValid;	// copy of events[nEvents-1] in scalar registers: most reports only touch the tail
	const unsigned char* doc;
	u32 docLen;
#ifdef SPA_PROF
	u64 prof[4];
#endif
};

__device__ __forceinline__ int ctxAt( const L1Params& P, const unsigned char* doc, u32 len, long pos)
{
	if (pos < 0 || pos >= (long)len) return CTX_EDGE;
	return P.classCtx[ P.byteClass[ doc[ pos]]];
}

// ---------------------------------------------------------------- stage 2: leftmost start per report
// lane-parallel: lane i resolves report base+i
struct LaneReport { u32 to, from, id, levelBind, prefixLen, suffixLen; };	// one queued report per lane, pattern attributes attached

template <bool LDS>
__device__ void resolveStarts( LexWave& w, const L1Params& P, const LexTab<LDS>& T, u32 base, u32 count, LaneReport& out)
{
	u32 i = base + LANE;
	out.to = 0; out.from = 0; out.id = 0; out.levelBind = 0; out.prefixLen = 0; out.suffixLen = 0;
	if (LANE < count)
	{
		const uint4 q = *(const uint4*)(w.queue + 4*(u64)i);	// {to, pattern, accLo|from, accHi}
		u32 to = q.x, pi = q.y & ~L1_LITERAL_FLAG;
		const uint4 p0 = *(const uint4*)&P.patterns[ pi], p1 = *((const uint4*)&P.patterns[ pi] + 1);	// {id,word,levelBind,prefixLen} {suffixLen,maskLo,maskHi,-}
		out.to = to; out.id = p0.x; out.levelBind = p0.z; out.prefixLen = p0.w; out.suffixLen = p1.x;
		if (q.y & L1_LITERAL_FLAG) { out.from = q.z; return; }		// whole-word literal: start already known
		u64 R = ((u64)q.w << 32) | q.z;
		const u32 pass = p0.y >> 6, ln = p0.y & 63u;
		const u64 mask = ((u64)p1.z << 32) | p1.y;
		const u64 shiftDst = T.at( T.oShift + pass*64 + ln), selfLoop = T.at( T.oSelf + pass*64 + ln);
		const u32 nEx = P.exCount[ pass];
		u32 from = to;
		long j = (long)to;			// R = positions that consumed byte j-1
		while (R && j > 0)
		{
			int prevctx = ctxAt( P, w.doc, w.docLen, j-2);
			if (R & T.at( T.oStart + (pass*CTX_COUNT + prevctx)*64 + ln)) from = (u32)(j-1);
			if (j-1 == 0) break;
			u64 Rp = ((R & shiftDst) >> 1) | (R & selfLoop);
			for (u32 e=0; e<nEx; ++e)
			{
				const u32 at = (pass*P.maxExceptions + e)*64 + ln;
				const u64 es = T.at( T.oExSrc + at), ed = T.at( T.oExDst + at);
				Rp |= (R & ed) ? es : 0ull;
			}
			u32 cls = P.byteClass[ w.doc[ j-2]];
			R = Rp & mask & T.at( (pass*P.nofClasses + cls)*64 + ln);
			--j;
		}
		out.from = from;
	}
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
__device__ u32 lookupSymbol( const LexWave& w, const L1Params& P, u32 lexemId, u32 from, u32 len)
{
	u32 h = 2166136261u;
	for (u32 b=0; b<4; ++b) h = symbolHashStep( h, (lexemId >> (8*b)) & 0xFFu);
	for (u32 k=0; k<len; ++k) h = symbolHashStep( h, uni( w.doc[ from+k]));
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
			for (u32 k=0; k<len && same; ++k) same = (uni( P.symbolText[ off+k]) == uni( w.doc[ from+k]));
			if (same) return ldu( &s->symbolId);
		}
		slot = (slot+1) & P.symbolMask;
	}
	return 0;
}

__device__ void handleReport( LexWave& w, const L1Params& P, u32 id, u32 lb, u32 pre, u32 suf, u32 from, u32 to)
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
		u32 sym = lookupSymbol( w, P, id, from, to-from);
		if (sym) patternid = sym;
	}
	Event ev; ev.id = id; ev.origpos = from; ev.origsize = to-from; ev.levelBind = lb & 0xFFFFu;
	const u32 level = lb & 0xFFu;
	const bool twin = (patternid != id);
	u32 n = w.nEvents;
	if (n + 2 > P.eventCap) { w.err = L1D_ERR_ARENA; return; }
	if (n && !w.tailValid) { ldEvent( w.tail, &w.events[ n-1]); w.tailValid = true; }
	if (n == 0)
	{
		stEvent( &w.events[0], ev); n = 1; w.tail = ev;
		if (twin) { Event t = ev; t.id = patternid; stEvent( &w.events[1], t); n = 2; w.tail = t; }
		w.nEvents = n; w.tailValid = true;
		return;
	}
	const u32 n0 = n;
	// delete pass (:757-777): from the back while origpos >= the new origpos
	const u32 lastPos = from + (to-from);
	u32 nofDeletes = 0;
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
			if ((m.levelBind & 0xFFu) > level && m.origpos <= ev.origpos) { w.nEvents = n; return; }
		}
	}
	// insert (:793-822), literal element moves of the reference (see oracle/l1_oracle.cpp for the
	// two-slot quirk of the symbol twin variant)
	const u32 newSlots = twin ? 2u : 1u;
	if (!nofDeletes && w.tailValid && !(w.tail.origpos > ev.origpos))
	{
		// common case: the new event goes behind the current tail, nothing moves
		stEvent( &w.events[ n], ev); w.tail = ev;
		if (twin) { Event t = ev; t.id = patternid; stEvent( &w.events[ n+1], t); w.tail = t; }
		w.nEvents = n + newSlots;
		return;
	}
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
	w.nEvents = n + newSlots;
	w.tailValid = false;		// reloaded on the next report
}

// drain the report queue: SOM in batches of 64 lanes, handler in report order
template <bool LDS>
__device__ void drainQueue( LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	for (u32 base=0; base<w.nQueue && !w.err; base+=64)
	{
		u32 count = w.nQueue - base < 64 ? w.nQueue - base : 64;
		u64 t0 = PROF_T();
		LaneReport lr;
		resolveStarts( w, P, T, base, count, lr);
		for (u32 k=0; k<count && !w.err; ++k)
		{
			handleReport( w, P, (u32)__builtin_amdgcn_readlane( lr.id, k), (u32)__builtin_amdgcn_readlane( lr.levelBind, k),
					(u32)__builtin_amdgcn_readlane( lr.prefixLen, k), (u32)__builtin_amdgcn_readlane( lr.suffixLen, k),
					(u32)__builtin_amdgcn_readlane( lr.from, k), (u32)__builtin_amdgcn_readlane( lr.to, k));
		}
		PROF_ACC( 1, t0);
	}
	w.nQueue = 0;
}


// ---------------------------------------------------------------- whole-word literals (\bWORD\b patterns)
// A maximal run of word characters [from,to) has ended: if it equals a literal, every pattern defined
// by that literal reports (from,to).  The reports are merged into the current end-offset group of the
// queue in ascending pattern index (the order the automaton reports have there already).
__device__ void literalReports( LexWave& w, const L1Params& P, u32 groupStart, u32 from, u32 to, u32 hash)
{
	const u32 len = to - from;
	u32 h = hash ? hash : 1u;
	u32 slot = h & P.literalMask;
	for (u32 probes=0; probes<=P.literalMask; ++probes)
	{
		// one 32-byte entry = one load: lane k < 8 fetches word k
		const u32 ew = (LANE < 8u) ? ((const u32*)&P.literals[ slot])[ LANE] : 0u;
		const u32 eh = (u32)__builtin_amdgcn_readlane( ew, 0);
		if (!eh) return;
		if (eh == h && (u32)__builtin_amdgcn_readlane( ew, 2) == len)
		{
			const u32 off = (u32)__builtin_amdgcn_readlane( ew, 1);
			const bool differ = LANE < len && w.doc[ from + LANE] != P.literalText[ off + LANE];	// one byte per lane (len <= 64)
			if (!__ballot( differ))
			{
				const u32 pb = (u32)__builtin_amdgcn_readlane( ew, 3), pc = (u32)__builtin_amdgcn_readlane( ew, 4);
				for (u32 k=0; k<pc; ++k)
				{
					const u32 pi = ldu( &P.litPats[ pb+k]);
					if (w.nQueue + 1 > P.queueCap) { w.err = L1D_ERR_ARENA; return; }
					u32 at = groupStart;
					while (at < w.nQueue && (ldu( &w.queue[ 4*(u64)at+1]) & ~L1_LITERAL_FLAG) < pi) ++at;
					for (u32 m=w.nQueue; m>at; --m)
					{
						u32* dst = w.queue + 4*(u64)m; const u32* src = w.queue + 4*(u64)(m-1);
						u32 a0 = ldu( &src[0]), a1 = ldu( &src[1]), a2 = ldu( &src[2]), a3 = ldu( &src[3]);
						dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
					}
					u32* q = w.queue + 4*(u64)at;
					q[0] = to; q[1] = pi | L1_LITERAL_FLAG; q[2] = from; q[3] = 0;
					w.nQueue += 1;
				}
				return;
			}
		}
		slot = (slot+1) & P.literalMask;
	}
}

// ---------------------------------------------------------------- stage 1: forward scan
template <int PASSES, bool LDS>
__device__ void scanDocument( LexWave& w, const L1Params& P, const LexTab<LDS>& T)
{
	u64 state[ PASSES];
#pragma unroll
	for (int p=0; p<PASSES; ++p) state[ p] = 0;
	const u32 len = w.docLen;
	// instances for 1..8 passes run tables of exactly that many passes; the 16 / 32 instances run anything up to it
	const u32 nofPasses = (PASSES <= 8) ? (u32)PASSES : uni( P.nofPasses);
	const u32 drainAt = P.queueCap > P.nofPatterns + 64 ? P.queueCap - P.nofPatterns - 64 : 0;
	int prevctx = CTX_EDGE;
	bool inWord = false; u32 runStart = 0, runHash = 0;		// token hash of the current run of word characters
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
	for (u32 tile=0; tile<=len && !w.err; tile+=64)
	{
		// 64 document bytes per load, one per lane; replayed byte by byte through a scalar register
		u32 mine = (tile + LANE < len) ? w.doc[ tile + LANE] : 0u;
		u32 inTile = (len - tile) < 64 ? (len - tile) : 64;	// document bytes in this tile
		// one byte step; the virtual step behind the last byte (matches that end with the document) is a
		// separate instance so that the hot one carries no end-of-document conditions
		auto step = [&]( auto atEndTag, const u32 k)
		{
			constexpr bool atEnd = decltype(atEndTag)::value;
			const u32 i = tile + k;
			const u32 b = atEnd ? 0u : (u32)__builtin_amdgcn_readlane( mine, k);
			const u32 sh = (b & 3u)*8;
			const u32 cls = atEnd ? 0u : (((u32)__builtin_amdgcn_readlane( clsReg, b >> 2) >> sh) & 0xFFu);
			const int ctx = atEnd ? (int)CTX_EDGE : (int)(((u32)__builtin_amdgcn_readlane( ctxReg, b >> 2) >> sh) & 0xFFu);
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
					cmRow[ p] = T.at( (p*P.nofClasses + cls)*64 + LANE);
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
					u64 nxt = ((st << 1) & shiftDst[ p]) | (st & selfLoop[ p]) | stRow[ p];
					const u32 nEx = nExOf[ p];
					for (u32 e=0; e<nEx; ++e)
					{
						const u32 at = (p*P.maxExceptions + e)*64 + LANE;
						const u64 es = T.at( T.oExSrc + at), ed = T.at( T.oExDst + at);
						nxt |= (st & es) ? ed : 0ull;
					}
					state[ p] = nxt & cmRow[ p];
				}
			}
			if (__ballot( anyAcc != 0))
			{
				u64 tHit = PROF_T();
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
					if (w.nQueue + total > P.queueCap) { w.err = L1D_ERR_ARENA; break; }
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
				PROF_ACC( 2, tHit);
			}
			if (P.nofLiterals)
			{
				const bool isW = (ctx == CTX_WORD);
				if (inWord && !isW)
				{
					inWord = false;
					if (i - runStart <= 64u) { u64 t0 = PROF_T(); __builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront"); literalReports( w, P, groupStart, runStart, i, runHash); PROF_ACC( 3, t0); }
				}
				if (isW)
				{
					if (!inWord) { inWord = true; runStart = i; runHash = 2166136261u; }
					runHash = symbolHashStep( runHash, b);
				}
			}
			prevctx = ctx;
			if (w.nQueue >= drainAt && w.nQueue)
			{
				__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
				drainQueue( w, P, T);
			}
		};
		for (u32 k=0; k<inTile && !w.err; ++k) step( std::false_type(), k);
		if (tile + 64 > len && !w.err) step( std::true_type(), inTile);
	}
	if (!w.err && w.nQueue)
	{
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		drainQueue( w, P, T);
	}
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

template <int PASSES, bool LDS>
__device__ void lexDocuments( const L1Params& P)
{
	LexTab<LDS> T;
	T.g = P.tableImage; T.oAccept = P.ldsAccept; T.oStart = P.ldsStart; T.oShift = P.ldsShift; T.oSelf = P.ldsSelf;
	T.oExSrc = P.ldsExSrc; T.oExDst = P.ldsExDst;
	if (LDS)
	{
		// the workgroup stages the hot tables once
		for (u32 k=threadIdx.x; k<P.ldsWords; k+=blockDim.x) ldsImage[ k] = P.tableImage[ k];
		__syncthreads();
	}
	const u32 waveSlot = uni( blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
	const u32 nWaveSlots = gridDim.x * (blockDim.x >> 6);
	u32* A = P.arenaBase + (u64)waveSlot * P.arenaWords;
	LexWave w;
	w.queue = A;
	w.events = (Event*)(A + 4*(u64)P.queueCap);
	// every wave starts with document `waveSlot` and takes its next ones from a device-side cursor;
	// the loop is bounded so that it ends whatever the cursor holds
	for (u32 round=0; round<=P.ndocs; ++round)
	{
		u32 doc = waveSlot;
		if (round)
		{
			u32 nx = 0;
			if (LANE == 0) nx = atomicAdd( (u32*)&P.counters[ L1C_CURSOR], 1u);
			doc = nWaveSlots + uni( nx);
		}
		if (doc >= P.ndocs) break;
		const u64 beg = ((u64)ldu( (const u32*)&P.docOffsets[ doc]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc]);
		const u64 end = ((u64)ldu( (const u32*)&P.docOffsets[ doc+1]+1) << 32) | ldu( (const u32*)&P.docOffsets[ doc+1]);
		w.doc = P.text + beg; w.docLen = (u32)(end - beg);
		w.nQueue = 0; w.nEvents = 0; w.err = 0; w.tailValid = false;
#ifdef SPA_PROF
		w.prof[0] = w.prof[1] = w.prof[2] = w.prof[3] = 0;
#endif
		{ u64 t0 = PROF_T(); scanDocument<PASSES,LDS>( w, P, T); PROF_ACC( 0, t0); }
		__builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront");
		if (!w.err) emitLexems( w, P, doc);
		else if (LANE == 0) { P.docRange[ 2*(u64)doc] = 0; P.docRange[ 2*(u64)doc+1] = 0; }
		if (LANE == 0)
		{
			P.docStatus[ doc] = (int32_t)w.err;
			atomicAdd( (unsigned long long*)&P.counters[ L1C_BYTES], (unsigned long long)w.docLen);
			if (w.err) atomicAdd( (unsigned long long*)&P.counters[ L1C_FAILED], 1ull);
#ifdef SPA_PROF
			for (int pi=0; pi<4; ++pi) atomicAdd( (unsigned long long*)&P.counters[ 4+pi], (unsigned long long)w.prof[ pi]);
#endif
		}
	}
}

} // anonymous namespace

// One instance per pass count (the per-pass rows of a byte step live in registers, so the count is a
// template parameter: an instance for 8 passes running a 6-pass table would spill three times as many
// registers).  Up to 5 passes fit the 128 registers of a 1024-thread workgroup without spills; 6..8
// spill 22..96 of them, which was measured faster than halving the waves (512-thread groups).
#define SPA_L1_KERNEL( NAME, N, T) \
extern "C" __global__ __launch_bounds__(T) void spa_l1_lex_kernel_##NAME( L1Params P) { if (P.ldsWords) lexDocuments<N,true>( P); else lexDocuments<N,false>( P); }
// up to 2 passes: under 96 registers (5 waves per SIMD, 20 per CU)
#define SPA_L1_KERNEL5( NAME, N, T) \
extern "C" __global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(5,8))) void spa_l1_lex_kernel_##NAME( L1Params P) { if (P.ldsWords) lexDocuments<N,true>( P); else lexDocuments<N,false>( P); }
SPA_L1_KERNEL5( p1, 1, 1024)
SPA_L1_KERNEL5( p2, 2, 1024)
SPA_L1_KERNEL( p3, 3, 1024)
SPA_L1_KERNEL( p4, 4, 1024)
SPA_L1_KERNEL( p5, 5, 1024)
SPA_L1_KERNEL( p6, 6, 1024)
SPA_L1_KERNEL( p7, 7, 1024)
SPA_L1_KERNEL( p8, 8, 1024)
SPA_L1_KERNEL( p16, 16, 256)
SPA_L1_KERNEL( p32, 32, 256)

namespace spa {
hipError_t launchL1Lex( const L1Params& P, unsigned nblocks, unsigned nthreads, hipStream_t stream)
{
	const size_t lds = (size_t)P.ldsWords * 8;
#define SPA_L1_LAUNCH( N) do { \
	if (lds > 65536) { hipError_t e = hipFuncSetAttribute( (const void*)spa_l1_lex_kernel_##N, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; } \
	hipLaunchKernelGGL( spa_l1_lex_kernel_##N, dim3( nblocks), dim3( nthreads), lds, stream, P); } while (0)
	switch (P.nofPasses)
	{
		case 1: SPA_L1_LAUNCH( p1); break;
		case 2: SPA_L1_LAUNCH( p2); break;
		case 3: SPA_L1_LAUNCH( p3); break;
		case 4: SPA_L1_LAUNCH( p4); break;
		case 5: SPA_L1_LAUNCH( p5); break;
		case 6: SPA_L1_LAUNCH( p6); break;
		case 7: SPA_L1_LAUNCH( p7); break;
		case 8: SPA_L1_LAUNCH( p8); break;
		default:
			if (P.nofPasses <= 16) SPA_L1_LAUNCH( p16);
			else if (P.nofPasses <= 32) SPA_L1_LAUNCH( p32);
			else return hipErrorInvalidValue;
	}
	return hipGetLastError();
}
}

