// Launch parameters of the level-2 kernel (shared by l2_kernel.hip and capi.cpp).
#ifndef SPA_L2_DEVICE_H
#define SPA_L2_DEVICE_H
#include <stdint.h>
#include "l2_tables.h"

namespace spa {

// per-document status codes (= SP_DOC_* of include/strus_pattern_amd.h)
enum {SPD_OK=0, SPD_ERR_ORDER=1, SPD_ERR_ARENA=2, SPD_ERR_KEYTRIGGERS=3, SPD_ERR_PASTFOLLOW=4, SPD_ERR_RANGE=5, SPD_ERR_DATAREF=6, SPD_ERR_LEXEMSIZE=7, SPD_ERR_INTERNAL=8, SPD_ERR_OUTPUT=9};
// counters[]
enum {SPC_RESULTS=0, SPC_ITEMS=1, SPC_EVENTS=2, SPC_FAILED=3, SPC_HANDOVER=4 /*documents the fast tier handed to the general kernel*/, SPC_COUNT=8};

// Per-wave arena: mutable state of the document a wavefront is working on.  Capacities in
// records, offsets in u32 words from the arena base.  Record sizes: rule 12 words, trigger 8,
// item 12, follow 12, stop-log 12, staged result 8, data reference 2, heap entry 2.
struct ArenaLayout
{
	uint32_t maxRules, maxTrigs, bucketCap, maxItems, maxRefs, maxFollow, maxDispose, maxHeap, maxGStack, maxStaged, nStop, winCap, scratchCap, winChunk, winChunks;
	uint32_t oRules, oTrigs, oBEvent, oBIdx, oBSize, oWindow, oHeap, oFollow, oDispose, oStop, oItems, oRefs, oGStack, oStaged, oRuleFree, oTrigFree, oItemFree, oRefFree, oWinArr, oScratch, oWinChunk, oWinFree;
	uint32_t totalWords;
};

struct L2Params
{
	// compiled tables (read only)
	const DevProgram* programs;
	const DevTrigDef* trigdefs;
	const DevKeyEntry* keytab;
	const DevKeyRef* keylist;
	uint32_t keymask;
	uint32_t nofStopWords;
	// input
	const uint32_t* lexems;		// sp_lexem_t[]: id, ordpos, origpos, origsize
	const uint32_t* origseg;	// optional
	const uint64_t* docOffsets;	// ndocs+1 lexem indices, or NULL when docRangesIn is given
	const uint64_t* docRangesIn;	// ndocs x (first lexem, count): the lexer kernel's output layout
	uint32_t ndocs;
	uint32_t withItems;
	// working memory
	uint32_t* arenaBase;
	ArenaLayout arena;
	uint32_t* docCursor;
	// output
	uint64_t* counters;		// SPC_*
	uint32_t* results;		// sp_result_t[resultCapacity] (9 words each)
	uint64_t resultCapacity;
	uint32_t* items;		// sp_result_item_t[itemCapacity] (7 words each)
	uint64_t itemCapacity;
	uint64_t* docRange;		// ndocs x (first result, count)
	uint64_t* docStats;		// ndocs x 4
	int32_t* docStatus;		// ndocs
	// patterns with format strings only (withFormats != 0)
	uint32_t withFormats;
	uint32_t* resultFormat;		// [resultCapacity] format handle of the result (0 = none)
	uint32_t* itemFormat;		// [itemCapacity] x {format handle, records of the item's subtree that follow it}
	uint32_t* trace;		// debug builds only (host-mapped), else NULL
	// list mode: the documents docList[0 .. *docListCount) instead of 0..ndocs (documents the fast tier handed
	// over, l2_fast.h); the output counters continue where the fast kernel left them
	const uint32_t* docList;
	const uint32_t* docListCount;
};

} // namespace
#endif
