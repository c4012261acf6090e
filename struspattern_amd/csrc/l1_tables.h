// Level-1 device tables shared by the host regex compiler (l1_compile.cpp) and the HIP lexer kernel
// (l1_kernel.hip).
//
// Model: every regex is compiled to a Glushkov position automaton (one state bit per byte-consuming
// position).  A pattern occupies a contiguous bit range inside ONE 64-bit word; words are grouped
// into passes of 64 words = one word per lane of a wavefront.  All zero-width assertions
// (\b \B ^ $) are compiled away by splitting positions by the context class of their byte
// (word / other / newline), so that inside a pattern every follow edge is unconditional; only the
// start of a match depends on the byte before it and the end on the byte after it:
//     state' = ( (state<<1 & shiftDst) | (state & selfLoop) | exceptions(state) | start[ctx(prev)] ) & charMask[class(byte)]
//     report = state & accept[ctx(next)]
#ifndef SPA_L1_TABLES_H
#define SPA_L1_TABLES_H
#include <stdint.h>

namespace spa {

// context class of a byte as seen by the assertions
enum {CTX_WORD=0, CTX_OTHER=1, CTX_NEWLINE=2, CTX_EDGE=3, CTX_COUNT=4};	// EDGE = begin/end of document

enum {L1_WORDS_PER_PASS=64};
enum {L1_WORD_LITERAL=0xFFFFFFFFu};	// pattern \bWORD\b handled by the token hash instead of automaton bits
enum {L1_LITERAL_FLAG=0x80000000u,	// queue records: start offset already known
      L1_DEAD_FLAG=0x40000000u};	// ... a candidate of the words kernel that did not confirm: skipped

struct DevLexPattern		// 32 B, one per defineLexem call in definition order; an expression too wide for one 64-bit word
{				// is cut at an alternation into several entries (adjacent, same defIndex) whose reports are merged
	uint32_t id;		// lexem id reported
	uint32_t word;		// global word index (pass*64 + lane) holding the pattern's positions; L1_WORD_LITERAL = none
	uint32_t levelBind;	// level | posbind<<8 | hasSymbols<<16 | hasSubexpr<<17
	uint32_t prefixLen;	// sub-expression selection: bytes cut at the front ...
	uint32_t suffixLen;	// ... and at the back of the raw match (fixed-length context)
	uint32_t maskLo, maskHi;// bits of the word that belong to this pattern
	uint32_t defIndex;	// definition index (0-based)
};

// Approximate literal tables (a table with a `~N` expression, src/patternLexer.cpp:333-412): every expression is
// a plain literal, matched by the approximate-matching kernel on characters
enum {L1_APPROX_MAXCHARS=24, L1_APPROX_MAXDIST=3, L1_APPROX_MAXPATTERNS=32};
struct DevApproxPattern		// 128 B, index = definition index
{
	uint32_t id;
	uint32_t levelBind;	// level | posbind<<8
	uint32_t editdist;
	uint32_t len;		// characters
	uint32_t byteLen;	// UTF-8 bytes
	uint32_t _pad[3];
	uint32_t cp[ L1_APPROX_MAXCHARS];	// code points
};

// ALLOWEMPTY: an expression that matches the empty string reports (offset, offset) wherever its empty path holds and nothing
// longer of it ends
struct DevNullable		// 16 B
{
	uint32_t pattern;	// patterns[] entry (the expression fits one automaton word)
	uint32_t emptyOk;	// bit (prev*CTX_COUNT + next)
	uint32_t _pad[2];
};

struct DevSymbol		// 32 B, open addressing (linear probing), hash==0 = empty
{
	uint32_t hash;
	uint32_t lexemId;
	uint32_t textOffset;
	uint32_t len;
	uint32_t symbolId;
	uint32_t _pad[3];
};

struct DevLiteral		// 48 B, open addressing, hash==0 = empty: a whole-word literal and the patterns defined by it
{
	uint32_t hash;
	uint32_t len;
	uint32_t patBegin;	// litPats[patBegin .. patBegin+patCount): pattern indices, ascending
	uint32_t patCount;
	uint32_t pat0;		// the first pattern and what the handler needs of it: one probe answers the common case
	uint32_t id0;
	uint32_t levelBind0;
	uint32_t textOffset;	// into the literal text pool
	uint8_t text[ 16];	// the first 16 bytes of the word, zero padded
};

// WORD SHAPES (round 3).  An expression every match of which ENDS WHERE A RUN OF WORD CHARACTERS ENDS and is pinned by a few
// literal bytes at a fixed place of that run needs no automaton pass over the text: the lexer looks the bytes up where a run
// ends, and a candidate is confirmed -- and its leftmost start found -- by running the expression's automaton backwards from
// the run's end (the start-of-match walk every automaton report goes through anyway).  Three shapes, each with a proof that no
// match is missed (l1_compile.cpp, wordShapeOf):
//   PREFIX(o,k)  \b S1..So LIT E \b   all of it word characters: a match is a whole run, LIT = its bytes [o, o+k)
//   SUFFIX(k)    X LIT \b            LIT word characters: a match ends with a run whose last k bytes are LIT
//   PREVWORD     \b WORD SEP E \b     WORD, E word characters, SEP one byte that is none: a match ends with a run whose
//                                     predecessor, one byte before it, is the whole run WORD
// The expressions keep their automaton positions in the tables (for the backward walk) but in passes of their own behind the
// passes the scan kernel runs (LexTables::scanPasses).
enum {SHAPE_PREFIX=1, SHAPE_SUFFIX=2, SHAPE_PREVWORD=3, SHAPE_MAXVARIANTS=8};
struct DevShape			// 16 B, open addressing (linear probing), tag==0 = empty
{
	uint32_t tag;		// kind | offset << 2 | length << 4 (PREFIX / SUFFIX: length 2..4 bytes) ; kind | length << 8 (PREVWORD: length of the word)
	uint32_t key;		// the literal bytes, first byte lowest; PREVWORD: literalHashFinish( polynomial hash of the word)
	uint32_t patBegin;	// shapePats[patBegin .. patBegin+patCount): pattern indices, ascending
	uint32_t patCount;
};
// What the kernel probes is a compact form of that table (8 bytes per entry, small enough to sit in LDS beside the automaton
// tables): fingerprint of (tag, key) -- nonzero, unique among the table's keys for the salt the compiler found -- and
// count << 24 | (count == 1 ? the pattern : patBegin).  A fingerprint that matches by chance costs a walk that does not confirm.
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t shapeFingerprint( uint32_t tag, uint32_t key, uint32_t salt)
{
	uint32_t h = (key ^ salt) * 0x85EBCA6Bu + tag * 0xC2B2AE35u;
	h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
	return h ? h : 1u;
}
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t shapeSlotHash( uint32_t tag, uint32_t key)
{
	uint32_t h = key * 0x9E3779B1u ^ (tag * 0x85EBCA6Bu);
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
	return h;
}

// FNV-1a step used for the symbol hash (lexem id first, then the text bytes); 0 is reserved for
// "empty" so a zero hash is mapped to 1
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t symbolHashStep( uint32_t h, uint32_t byte) { return (h ^ byte) * 16777619u; }

// Hash of a whole-word literal: polynomial in the bytes, so that the kernel gets the hashes of all words of a
// 64-byte tile with one segmented scan over the lanes (h = h*MUL + byte+1 per byte, then a finishing mix
// because the low bits of a polynomial hash only see the low bits of the bytes); 0 is reserved for "empty".
enum {L1_LITHASH_MUL=0x9E3779B1u};
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t literalHashFinish( uint32_t h)
{
	h ^= h >> 16; h *= 0x7feb352dU;
	h ^= h >> 15; h *= 0x846ca68bU;
	h ^= h >> 16;
	return h ? h : 1u;
}

} // namespace
#endif
