// Launch parameters of the level-1 lexer kernel (shared by l1_kernel.hip and capi_l1.cpp).
#ifndef SPA_L1_DEVICE_H
#define SPA_L1_DEVICE_H
#include <stdint.h>
#include "l1_tables.h"

namespace spa {

enum {L1C_LEXEMS=0, L1C_BYTES=1, L1C_RAW=2, L1C_FAILED=3, L1C_COUNT=8, L1C_CURSOR=8 /*document cursor of the scan kernel, behind the counters*/, L1C_CURSOR2=9 /*of the post-processing kernel*/,
      L1C_CURSOR3=10 /*of the sequential re-scan*/, L1C_UNITS=11 /*scan units = chunks of all documents*/, L1C_CHUNKED=12 /*some document has more than one chunk*/, L1C_SEQDOCS=13 /*documents scanned again in one piece*/,
      L1C_CURSOR4=14 /*unit cursor of the words kernel*/, L1C_WORDREPORTS=15 /*records the words kernel wrote*/,
      L1C_OVER_QUEUE=16 /*units whose slice of a report queue was too small*/, L1C_OVER_EVENTS=17 /*documents whose event array was too small*/, L1C_ALLOC=18};

// words kernel: waves of a workgroup (12, or 16 while its table image leaves room), static LDS per wave (ring + run ends); the image takes the rest of the 160 KB
enum {L1_WORD_WAVES=12, L1_WORD_WAVES_SMALL=16, L1_WORDS_LDS_PER_WAVE=512*2 + 96*5*4};

struct L1Params
{
	// compiled tables (read only)
	const uint8_t* byteClass;	// 256
	const uint8_t* classCtx;	// nofClasses
	const uint64_t* charMask;	// [pass][class][64]
	const uint64_t* startMask;	// [pass][4][64]
	const uint64_t* acceptMask;	// [pass][4][64]
	const uint64_t* shiftDst;	// [pass][64]
	const uint64_t* selfLoop;	// [pass][64]
	const uint64_t* exSrc;		// [pass][maxExceptions][64]
	const uint64_t* exDst;
	const uint32_t* exCount;	// [pass]
	// Two unused slots, kept on purpose: with them the 3-pass instance runs 6% faster (115 vs 122 ms on half a
	// bench step, same box, three runs) -- the offsets of the fields behind them decide how the compiler groups
	// its scalar loads of the kernel arguments.  Remove them only with a measurement.
	const uint32_t* _layout0;
	const uint32_t* _layout1;
	const uint32_t* patOfBit;	// [word][64]: pattern owning automaton bit
	const DevLexPattern* patterns;
	const DevSymbol* symbols;
	const uint8_t* symbolText;
	uint32_t symbolMask;
	const DevLiteral* literals;	// whole-word literal hash table
	const uint8_t* literalText;
	const uint32_t* litPats;
	uint32_t literalMask, nofLiterals;
	const uint64_t* tableImage;	// [charMask][acceptMask][startMask][shiftDst][selfLoop][exSrc][exDst] back to back, for LDS staging
	uint32_t ldsWords;		// 0 = read the tables from global memory
	uint32_t ldsAccept, ldsStart, ldsShift, ldsSelf, ldsExSrc, ldsExDst;	// word offsets inside the image
	uint32_t nofPasses, nofClasses, maxExceptions, nofPatterns;
	uint32_t reportsOrdered;	// 0: the reports of one end offset have to be sorted by pattern index
	// input
	const uint8_t* text;		// all documents back to back
	const uint64_t* docOffsets;	// ndocs+1 byte offsets
	uint32_t ndocs;
	// per-wave working memory
	uint32_t* arenaBase;
	uint64_t arenaWords;		// words per wave
	uint32_t queueCap;		// (unused since the kernels were split)
	uint32_t eventCap;		// event array records (4 words each)
	// hand-over between the scan kernel and the post-processing kernel: the raw reports of document d, 16 B each,
	// at reportQueue[ 4*qbase(d) ..) with qbase(d) = (begin(d)*queueMul >> 4) + 64*d -- room for queueMul/16 reports
	// per text byte plus 64; reportCount[d] = how many there are
	uint32_t* reportQueue;
	uint32_t* reportCount;
	uint32_t queueMul;
	// output
	uint64_t* counters;		// L1C_*
	uint32_t* lexems;		// sp_lexem_t[lexemCapacity]
	uint64_t lexemCapacity;
	uint64_t* docRange;		// ndocs x (first lexem, count)
	int32_t* docStatus;		// ndocs
	// approximate literal tables (new fields go behind everything else: see the note on the layout slots above)
	const DevApproxPattern* approx;	// nofApprox patterns, or null
	uint32_t nofApprox;
	uint32_t* charCp;		// characters of document d: charCp / charPos [ begin(d)+d .. ), code point and byte offset
	uint32_t* charPos;
	const uint16_t* cpBlocks;	// classes by code point (LexTables::cpBlocks / cpPages), or null
	const uint8_t* cpPages;
	// documents longer than chunkBytes are scanned as several units (unit = one chunk of one document): unitStart[d] = first
	// unit of document d (ndocs+1 entries, written by the units kernel); the raw reports of unit u lie at
	// reportQueue[ 4*((byte offset of the chunk * queueMul >> 4) + 64*u) ..), reportCount[u] of them
	uint32_t* unitStart;
	uint32_t chunkBytes;		// multiple of 64
	uint32_t* docSequential;	// [ndocs]: 1 = a chunk's start state could not be proven from its warm-up; the document is scanned again in one piece
	uint32_t sequentialPass;	// this launch of the scan kernel is that re-scan
	const DevNullable* nullable;	// ALLOWEMPTY: expressions that match the empty string (or null)
	uint32_t nofNullable;
	uint32_t ucp;			// option UCP: contexts by Unicode word characters, byteClass has the 64 twin entries [256..319]
	uint32_t splitPatterns;		// some expression is cut into several patterns entries (same defIndex): their reports are merged
	// words kernel (round 3): whole-word literals and word shapes (l1_tables.h) are found where runs of word characters end, by a
	// kernel of its own with a lane per byte; its reports of unit u -- start already known, (end offset, pattern) order, records
	// of candidates that did not confirm marked L1_DEAD_FLAG -- lie at wordQueue[ 4*queueBase(u) ..), wordCount[u] of them
	const DevShape* shapes; const uint32_t* shapePats;
	uint32_t shapeMask, nofShapeVariants;
	uint32_t shapeVariants[ SHAPE_MAXVARIANTS];
	uint32_t* wordQueue;
	uint32_t* wordCount;
	uint32_t wordsKernel;		// 1: the post-processing kernel merges the two queues (plain tables); 0: it finds the literals itself
	uint32_t shapeFpOffset;		// the compact shape table (l1_tables.h) lies at this word offset of the table image (LDS or global, like the rest)
	uint32_t shapeSalt;
	uint32_t scanWords;		// automaton words the scanned passes use: up to 4 words in one pass take the lane-per-stream scan kernel
	uint32_t ldsChar;		// offset of the character rows in the image (0; the words kernel's image leaves the scanned passes out: biased, as the other offsets)
	uint32_t postClusters;		// 1: the handler runs a cluster of reports per lane (postDocumentClusters); 0: one report after the other
};

} // namespace
#endif
