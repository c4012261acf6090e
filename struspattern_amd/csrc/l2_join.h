// PROTOTYPE, opt-in (SPA_L2_JOIN=1): the rule automaton WITHOUT materialised rule instances, for rule sets made of two-term
// `sequence( A, B | range )` programs that nobody listens to ("The ceiling").
// The reference installs an instance at every A and retires it at the first later B or when it expires
// (src/ruleMatcherAutomaton.cpp:589-1334); the result SET of a document follows from the positions alone:
//   (A at lexem i, B at lexem j) matches  iff  ordpos(i) < ordpos(j) <= ordpos(i) + range  and no B lies between them at a
//   position behind i's (that B would have taken the instance).
// So every lexem j looks back over the lexems of the last `maxRange` positions and asks a hash table keyed by the PAIR of event
// ids (id(i), id(j)) for the programs it completes -- lane-parallel over the lexems, no per-document state at all (a first
// version that sorted the document's (id, index) pairs in LDS and searched them per rule was 7 x SLOWER than the exact
// engine: it pays per (lexem, rule that ends with it) pair, which is as many as the exact engine's installs).
// What this mode does NOT reproduce: the order of the results inside a document (the reference's depends on the swap history
// of its trigger buckets) and the statistics (nothing is installed); results come in (end lexem, start lexem descending,
// rule) order.  Parity is therefore checked on result SETS (tests/test_l2_join_gpu.py).
#ifndef SPA_L2_JOIN_H
#define SPA_L2_JOIN_H
#include <stdint.h>

namespace spa {


struct JoinKey			// 16 B, open addressing by joinHash( first, second), first==0 = empty: the programs sequence( first, second | .. )
{
	uint32_t first, second;
	uint32_t begin;		// rules[begin .. begin+count), definition order
	uint32_t count;
};
enum {JOIN_FILTER_WORDS=4096};		// 16 KB = 131072 bits
enum {JOIN_SELF=0xFFFFFFFFu};		// `first` of the entries for any( .. ): the lexem alone is the match
enum {JOIN_STRUCT=1u};			// JoinRule::flags: no delimiter lexem may lie between the two terms (*_struct);
					// bits 8..15 / 16..23: the variable attached to the first / the completing term (0 = none)
struct JoinRule			// 16 B
{
	uint32_t range;
	uint32_t resultHandle;
	uint32_t formatHandle;
	uint32_t flags;
};
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t joinHash( uint32_t first, uint32_t second)
{
	uint32_t h = first * 0x9E3779B1u ^ (second + 0x7F4A7C15u) * 0x85EBCA6Bu;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
	return h;
}

struct JoinParams
{
	const JoinKey* keytab; uint32_t keymask;
	const JoinRule* rules;
	const uint32_t* filter;		// JOIN_FILTER_WORDS words: bit (joinHash( first, second) >> 8) % bits set for every table key -- copied to LDS,
					// so that most pairs of nearby lexems (which complete nothing) are refused without a memory access
	uint32_t* counts;		// per lexem of the batch: its number of matches | items << 16 (first pass -> second pass)
	uint64_t countsCapacity;	// lexem indices below it have a slot (a document beyond it counts twice instead)
	uint32_t maxRange;		// the largest position range of any program
	uint32_t delimiter;		// the delimiter event of the *_struct programs (0 = none)
	const uint32_t* lexems;		// sp_lexem_t[]: id, ordpos, origpos, origsize
	const uint32_t* origseg;	// optional
	const uint64_t* docOffsets;	// ndocs+1 lexem indices, or NULL when docRangesIn is given
	const uint64_t* docRangesIn;	// ndocs x (first lexem, count)
	uint32_t ndocs;
	uint32_t* docCursor;
	uint64_t* counters;		// SPC_*
	uint32_t* results; uint64_t resultCapacity;
	uint32_t* items; uint64_t itemCapacity; uint32_t withItems; uint32_t* itemFormat;	// sp_result_item_t[] (7 words each)
	uint64_t* docRange; uint64_t* docStats; int32_t* docStatus;
	uint32_t withFormats; uint32_t* resultFormat;
};

} // namespace
#endif
