// PROTOTYPE of the non-materialising rule automaton (l2_join.h): lane-parallel over the lexems of a document, no state.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "l2_join.h"
#include "l2_tables.h"
#include "l2_device.h"
#include "wave_scan.h"

using namespace spa;
typedef uint32_t u32;
typedef uint64_t u64;

namespace {
#define LANE ((u32)(threadIdx.x & 63u))
__device__ __forceinline__ u32 uni( u32 v) { return __builtin_amdgcn_readfirstlane( v); }
__shared__ u32 pairFilter[ JOIN_FILTER_WORDS];
__device__ __forceinline__ bool maybeKey( u32 first, u32 second)
{
	const u32 bit = (joinHash( first, second) >> 8) % (JOIN_FILTER_WORDS*32u);
	return (pairFilter[ bit >> 5] >> (bit & 31u)) & 1u;
}

// the matches that end at lexem j (lane-private loop over the lexems of the last maxRange positions); WRITE: store them
// from `out` on, else count
// records are word aligned only: multi-word stores through memcpy (one request per 16 / 12 bytes instead of one per word --
// the kernel is bound by the number of store requests of its lanes, not by their bytes)
__device__ __forceinline__ void st4( u32* p, u32 a, u32 b, u32 c, u32 d) { const u32 v[ 4] = {a, b, c, d}; __builtin_memcpy( p, v, 16); }
__device__ __forceinline__ void st3( u32* p, u32 a, u32 b, u32 c) { const u32 v[ 3] = {a, b, c}; __builtin_memcpy( p, v, 12); }
__device__ __forceinline__ void putItem( const JoinParams& P, u64 at, u32 variable, uint4 t, u32 st)
{
	u32* io = P.items + at*7;
	st4( io, variable, t.y, t.y + 1u, st); st3( io+4, t.z, st, t.z + t.w);
	if (P.withFormats) { P.itemFormat[ 2*at] = 0; P.itemFormat[ 2*at+1] = 0; }
}
// returns matches | items << 16
template <bool WRITE>
__device__ __forceinline__ u32 matchesEndingAt( const JoinParams& P, const uint4* lex, const u32* seg, u32 j, uint4 lj, u32* out, u32* fmtOut, u64 itemAt)
{
	u32 cnt = 0, icnt = 0;
	const u32 e = lj.x;
	if (!e) return 0;
	// any( .., e, .. ): the lexem alone
	if (maybeKey( JOIN_SELF, e))
	{
		u32 slot = joinHash( JOIN_SELF, e) & P.keymask;
		for (u32 probes=0; probes<=P.keymask; ++probes)
		{
			const JoinKey k = P.keytab[ slot];
			if (!k.first) break;
			if (k.first == (u32)JOIN_SELF && k.second == e)
			{
				for (u32 r=0; r<k.count; ++r)
				{
					const JoinRule rule = P.rules[ k.begin + r];
					// (any( e, e ): both triggers of the instance take the lexem, the second field is the one that fired first)
					const u32 vj = P.withItems ? (rule.flags >> 16) & 0xFFu : 0u, vi = P.withItems ? (rule.flags >> 8) & 0xFFu : 0u;
					const u32 ni = (vj ? 1u : 0u) + (vi ? 1u : 0u);
					if (WRITE)
					{
						u32* o = out + 9*(u64)cnt;
						const u32 sj = seg ? seg[ j] : 0u;
						st4( o, rule.resultHandle, lj.y, lj.y + 1u, sj); st4( o+4, lj.z, sj, lj.z + lj.w, P.withItems ? (u32)(itemAt + icnt) : 0u); o[8] = ni;
						if (fmtOut) fmtOut[ cnt] = rule.formatHandle;
						if (vj) putItem( P, itemAt + icnt, vj, lj, sj);
						if (vi) putItem( P, itemAt + icnt + (vj ? 1u : 0u), vi, lj, sj);
					}
					++cnt; icnt += ni;
				}
				break;
			}
			slot = (slot+1) & P.keymask;
		}
	}
	u32 takenPos = 0;		// position of the latest occurrence of e seen so far: it has taken every instance that began at an earlier position
	bool delimited = false;		// a delimiter lexem lies between i and j
	for (u32 i=j; i-- > 0;)
	{
		const uint4 li = lex[ i];
		if (lj.y - li.y > P.maxRange) break;		// every instance that old has expired
		if (takenPos > li.y) break;			// this one and all earlier ones were completed by that occurrence
		if (li.x && lj.y > li.y && maybeKey( li.x, e))
		{
			u32 slot = joinHash( li.x, e) & P.keymask;
			for (u32 probes=0; probes<=P.keymask; ++probes)
			{
				const JoinKey k = P.keytab[ slot];
				if (!k.first) break;
				if (k.first == li.x && k.second == e)
				{
					for (u32 r=0; r<k.count; ++r)
					{
						const JoinRule rule = P.rules[ k.begin + r];
						if (lj.y - li.y > rule.range || (delimited && (rule.flags & JOIN_STRUCT))) continue;
						// captured items, latest first (as the reference lists them): the completing lexem's, then the first one's
						const u32 vi = P.withItems ? (rule.flags >> 8) & 0xFFu : 0u, vj = P.withItems ? (rule.flags >> 16) & 0xFFu : 0u;
						const u32 ni = (vi ? 1u : 0u) + (vj ? 1u : 0u);
						if (WRITE)
						{
							u32* o = out + 9*(u64)cnt;
							const u32 si = seg ? seg[ i] : 0u, sj = seg ? seg[ j] : 0u;
							st4( o, rule.resultHandle, li.y, lj.y + 1u, si); st4( o+4, li.z, sj, lj.z + lj.w, P.withItems ? (u32)(itemAt + icnt) : 0u); o[8] = ni;
							if (fmtOut) fmtOut[ cnt] = rule.formatHandle;
							if (vj) putItem( P, itemAt + icnt, vj, lj, sj);
							if (vi) putItem( P, itemAt + icnt + (vj ? 1u : 0u), vi, li, si);
						}
						++cnt; icnt += ni;
					}
					break;
				}
				slot = (slot+1) & P.keymask;
			}
		}
		if (li.x == e && li.y > takenPos) takenPos = li.y;
		if (P.delimiter && li.x == P.delimiter) delimited = true;
	}
	return cnt | (icnt << 16);
}

__device__ void joinDocuments( const JoinParams& P)
{
	for (u32 k=threadIdx.x; k<(u32)JOIN_FILTER_WORDS; k+=blockDim.x) pairFilter[ k] = P.filter[ k];
	__syncthreads();
	for (u32 round=0; round<=P.ndocs; ++round)
	{
		u32 doc = 0;
		if (LANE == 0) doc = atomicAdd( P.docCursor, 1u);
		doc = uni( doc);
		if (doc >= P.ndocs) break;
		u64 beg, n64;
		if (P.docRangesIn) { beg = P.docRangesIn[ 2*(u64)doc]; n64 = P.docRangesIn[ 2*(u64)doc+1]; }
		else { beg = P.docOffsets[ doc]; n64 = P.docOffsets[ doc+1] - beg; }
		beg = ((u64)uni( (u32)(beg >> 32)) << 32) | uni( (u32)beg);
		const u32 n = uni( (u32)n64);
		const uint4* lex = (const uint4*)P.lexems + beg;
		const u32* seg = P.origseg ? P.origseg + beg : 0;
		u32 err = 0;
		if (n64 >= (1ull << 32)) err = SPD_ERR_RANGE;
		const bool stored = beg + n64 <= P.countsCapacity;
		// checks of putInput (patternMatcher.cpp:131-162), the matches counted
		u32 total = 0, itotal = 0;
		if (!err)
		{
			bool bad = false, order = false;
			for (u32 base=0; base<n; base+=64)
			{
				const u32 j = base + LANE;
				u32 c = 0;
				if (j < n)
				{
					const uint4 lj = lex[ j];
					if (lj.x >= (1u<<29) || lj.z >= 0x7FFFFFFFu || lj.w >= 0x7FFFFFFFu) bad = true;
					if (j + 1 < n && lex[ j+1].y < lj.y) order = true;
					c = matchesEndingAt<false>( P, lex, seg, j, lj, 0, 0, 0);
					if ((c & 0xFFFFu) > 0x7FFFu) bad = true;		// (two items per match at most: both halves fit 16 bits)
					if (stored) P.counts[ beg + j] = c;
				}
				const u32 incl = waveScanAdd( c & 0xFFFFu), iincl = waveScanAdd( c >> 16);
				total += uni( (u32)__shfl( (int)incl, 63));
				itotal += uni( (u32)__shfl( (int)iincl, 63));
			}
			if (__ballot( order)) err = SPD_ERR_ORDER; else if (__ballot( bad)) err = SPD_ERR_RANGE;
		}
		u64 resBase = 0;
		if (!err && total)
		{
			u64 b = 0;
			if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_RESULTS], (unsigned long long)total);
			resBase = ((u64)uni( (u32)(b >> 32)) << 32) | uni( (u32)b);
			if (resBase + total > P.resultCapacity) { err = SPD_ERR_OUTPUT; total = 0; }
		}
		u64 itemBase = 0;
		if (!err && total && itotal)
		{
			u64 b = 0;
			if (LANE == 0) b = atomicAdd( (unsigned long long*)&P.counters[ SPC_ITEMS], (unsigned long long)itotal);
			itemBase = ((u64)uni( (u32)(b >> 32)) << 32) | uni( (u32)b);
			if (itemBase + itotal > P.itemCapacity) { err = SPD_ERR_OUTPUT; total = 0; }
		}
		if (!err && total)
		{
			u32 at = 0, iat = 0;
			for (u32 base=0; base<n; base+=64)
			{
				const u32 j = base + LANE;
				uint4 lj = make_uint4( 0, 0, 0, 0);
				u32 c = 0;
				if (j < n)
				{
					if (stored) { c = P.counts[ beg + j]; if (c) lj = lex[ j]; }
					else { lj = lex[ j]; c = matchesEndingAt<false>( P, lex, seg, j, lj, 0, 0, 0); }
				}
				const u32 cm = c & 0xFFFFu, ci = c >> 16;
				const u32 incl = waveScanAdd( cm), iincl = waveScanAdd( ci);
				if (cm)
				{
					const u64 mine = resBase + at + incl - cm;
					(void)matchesEndingAt<true>( P, lex, seg, j, lj, P.results + 9*mine, P.withFormats ? P.resultFormat + mine : 0, itemBase + iat + iincl - ci);
				}
				at += uni( (u32)__shfl( (int)incl, 63));
				iat += uni( (u32)__shfl( (int)iincl, 63));
			}
		}
		if (LANE == 0)
		{
			P.docRange[ 2*(u64)doc] = resBase; P.docRange[ 2*(u64)doc+1] = err ? 0 : total;
			u64* st = P.docStats + 4*(u64)doc;
			st[0] = 0; st[1] = 0; st[2] = 0; st[3] = 0;
			P.docStatus[ doc] = (int32_t)err;
			atomicAdd( (unsigned long long*)&P.counters[ SPC_EVENTS], (unsigned long long)(err ? 0 : n));
			if (err) atomicAdd( (unsigned long long*)&P.counters[ SPC_FAILED], 1ull);
		}
	}
}
} // anonymous namespace

extern "C" __global__ __launch_bounds__(256) void spa_l2_join_kernel( JoinParams P) { joinDocuments( P); }

namespace spa {
hipError_t launchL2Join( const JoinParams& P, unsigned nwaves, hipStream_t stream)
{
	hipLaunchKernelGGL( spa_l2_join_kernel, dim3( (nwaves + 3) / 4), dim3( 256), 0, stream, P);
	return hipGetLastError();
}
}
