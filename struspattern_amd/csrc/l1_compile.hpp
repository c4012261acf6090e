// Level-1 lexer compiler (host side of the product): regular expressions -> bit-parallel position
// automaton tables for the HIP lexer kernel.  Replaces PatternTable + hs_compile_ext_multi of the
// reference (src/patternLexer.cpp:231-649, :1068-1118); Hyperscan itself is not used.
#ifndef SPA_L1_COMPILE_HPP
#define SPA_L1_COMPILE_HPP
#include <stdint.h>
#include <cstddef>
#include <map>
#include <string>
#include <vector>
#include "l1_tables.h"

namespace spa {

enum {LEX_CASELESS=1, LEX_DOTALL=2, LEX_MULTILINE=4, LEX_ALLOWEMPTY=8, LEX_UCP=16, LEX_BYTECHAR=32};

struct LexTables
{
	uint32_t nofPasses;
	uint32_t nofClasses;
	uint32_t maxExceptions;			// per word, over all passes
	bool ucp;				// option UCP: contexts (word / not) and \w \d \s by Unicode properties
	std::vector<uint8_t> byteClass;		// 256 -> class id; with UCP 320: [256..319] = continuation bytes 80..BF of a word character
	std::vector<uint8_t> classCtx;		// class id -> CTX_*
	// classes by code point for the lead byte of a well-formed multi-byte character (only when some expression holds
	// a large code point set, e.g. \p{Lu}): class = cpPages[ cpBlocks[ cp>>6]*64 + (cp&63)], 0xFF = none (use the byte's class)
	std::vector<uint16_t> cpBlocks;		// empty or 0x110000/64 entries
	std::vector<uint8_t> cpPages;
	std::vector<uint64_t> charMask;		// [pass][class][64]
	std::vector<uint64_t> startMask;	// [pass][CTX_COUNT][64]
	std::vector<uint64_t> acceptMask;	// [pass][CTX_COUNT][64]
	std::vector<uint64_t> shiftDst;		// [pass][64]
	std::vector<uint64_t> selfLoop;		// [pass][64]
	std::vector<uint32_t> exCount;		// [pass]: exception entries used by the widest word of the pass
	std::vector<uint64_t> exSrc;		// [pass][maxExceptions][64]
	std::vector<uint64_t> exDst;		// [pass][maxExceptions][64]
	std::vector<uint32_t> wordPatBegin;	// [nofPasses*64+1] -> wordPats
	std::vector<uint32_t> wordPats;		// pattern indices (0-based) per word, ascending
	std::vector<uint32_t> patOfBit;		// [nofPasses*64][64]: pattern owning each automaton bit
	std::vector<DevLexPattern> patterns;
	std::vector<DevSymbol> symbols;		// power-of-two size (>=1)
	std::vector<uint8_t> symbolText;
	std::vector<DevLiteral> literals;	// power-of-two size (>=1)
	std::vector<uint8_t> literalText;
	std::vector<uint32_t> litPats;
	std::vector<DevNullable> nullable;	// ALLOWEMPTY: the expressions that match the empty string (at most 64)
	std::vector<DevApproxPattern> approx;	// non-empty: approximate literal table, the automaton tables are empty
	// word shapes (l1_tables.h): expressions found at the ends of word runs instead of by an automaton pass
	std::vector<DevShape> shapes;		// power-of-two size (>=1)
	std::vector<uint32_t> shapePats;
	std::vector<uint64_t> shapeFp;		// compact form the kernel probes (l1_tables.h): same slots as `shapes`
	uint32_t shapeSalt;
	std::vector<uint32_t> shapeVariants;	// the distinct tags (kind | offset << 2 | length << 4; PREVWORD: kind) the table holds
	uint32_t nofShapes;			// expressions taken as shapes
	uint32_t scanPasses;			// passes [0, scanPasses) are run by the scan kernel; the rest is only walked backwards
	uint32_t scanWords;			// automaton words the scanned passes use (<= scanPasses * 64)
	bool lanesOk;				// no scanned expression can stay live across a blank (a position on a cycle that takes ' '): the state at the
						// start of a piece of text is then known after a short warm-up, which the lane-per-stream scan kernel relies on
	uint32_t nofLiterals;
	uint32_t nofPositions;
	bool reportsOrdered;			// patterns sit in the words in definition order (else the kernel sorts the reports of one end offset)
};

class LexCompiler
{
public:
	LexCompiler() :m_options(0),m_compiled(false){}

	// PatternLexerInstanceInterface (src/patternLexer.cpp:971-1141); throw std::runtime_error
	void defineLexemName( uint32_t id, const std::string& name);
	const char* getLexemName( uint32_t id) const;
	void defineLexem( uint32_t id, const std::string& expression, uint32_t resultIndex, uint32_t level, int posbind);
	void defineSymbol( uint32_t symbolid, uint32_t patternid, const std::string& name);
	uint32_t getSymbol( uint32_t patternid, const std::string& name) const;
	void defineOption( const std::string& name, double value);
	size_t nofDefinitions() const { return m_defs.size(); }
	void compile();
	bool compiled() const			{return m_compiled;}
	const LexTables& tables() const		{return m_tables;}
	size_t nofPatterns() const		{return m_defs.size();}
	// compiled tables as a blob and back (serial.hpp): a loaded lexer is compiled and frozen
	void save( std::vector<uint8_t>& out) const;
	void load( const void* blob, size_t size);

private:
	struct Def
	{
		std::string expression;
		uint32_t id, resultIndex, level, editdist;
		int posbind;
	};
	std::vector<Def> m_defs;
	std::map<uint32_t, std::map<std::string,uint32_t> > m_symbols;
	std::map<uint32_t,std::string> m_names;
	unsigned m_options;
	bool m_compiled;
	LexTables m_tables;
};

} // namespace
#endif
