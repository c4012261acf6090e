// TEST INFRASTRUCTURE ONLY -- CPU oracle for level 1 (pattern lexer).
//
// The reference's level-1 arithmetic lives in Intel Hyperscan v5.1.1 (dist/travis/script.sh:130-133)
// and libtre, neither of which is in /root/reference or in this image.  This oracle therefore
// restates (a) Hyperscan's published block-mode semantics with HS_FLAG_SOM_LEFTMOST [|HS_FLAG_UTF8]
// as summarised in SURVEY.md App. A.2 -- for every pattern and every end offset at which a
// non-empty match ends: one report (pattern, leftmost start, end), ordered by end offset then
// pattern index -- and (b) literally the reference's own code after the scan:
// match_event_handler (src/patternLexer.cpp:717-826) and the ordinal-position pass (:893-945).
//
// PINNING: the only fixture that pins (a) against real Hyperscan is the 36-lexem vector of
// tests/charRegexMatch (testCharRegexMatch.cpp:107-159), checked in tests/test_oracle_l1.py.
// Beyond those lexems parity with Hyperscan is UNPINNED; the regex semantics are additionally
// cross-checked against Python's `re` on small random cases (same test file).
//
// EDIT DISTANCE (`expr ~N`, patternLexer.cpp:333-412, :414-426, :450-601): the reference pre-matches on a
// one-byte-per-character hash of text and expression (unicodeUtils.cpp:19-44) with Hyperscan's
// approximate matching and re-matches every candidate with libtre's approximate matcher.  Neither library
// is here; the restatement (l1_oracle.cpp, "approximate literal tables") covers tables whose expressions
// are all plain literals and is pinned by the two vectors of testCharRegexMatch.cpp:161-196 ONLY -- the
// tie-breaking of the second stage is a model fitted to those vectors, stated where it is implemented.
//
// Deliberately a different algorithm from the product (which builds a Glushkov position automaton
// with bit-parallel tables): here the regex AST is compiled to a Thompson-style epsilon NFA and
// simulated with a per-state minimal start offset.
#ifndef SPA_ORACLE_L1_HPP
#define SPA_ORACLE_L1_HPP
#include <stdint.h>
#include <cstddef>
#include <string>
#include <vector>
#include <map>
#include <stdexcept>

namespace oracle {

enum LexOption {OptCaseless=1, OptDotAll=2, OptMultiline=4, OptAllowEmpty=8, OptUcp=16, OptByteChar=32};
enum PosBind {BindContent=0, BindSuccessor=1, BindPredecessor=2, BindUnique=3};

struct RawMatch { uint32_t idx; uint32_t from; uint32_t to; };	// idx = 1-based definition index (patternLexer.cpp:390)
struct LexemOut { uint32_t id, ordpos, origpos, origsize; };

// One compiled regular expression (epsilon NFA over bytes)
class Regex
{
public:
	enum NodeType {Char, Eps, Split, AssertWB, AssertNWB, AssertBOL, AssertEOL, AssertBOD, AssertEOD, Accept};
	struct Node { NodeType type; int out, out1; uint32_t set[8]; };	// set: 256-bit byte set for Char

	Regex( const std::string& expr, unsigned options);
	// all (from,to) with from leftmost per `to`, non-empty, ascending `to`
	void scan( const unsigned char* src, size_t len, std::vector<std::pair<uint32_t,uint32_t> >& out) const;
	int start() const {return m_start;}
	const std::vector<Node>& nodes() const {return m_nodes;}
	// byte length range of the text before / after capture group `group` when every element around
	// it has a fixed length; returns false otherwise (used for resultIndex, see l1_oracle.cpp)
	bool fixedContext( unsigned group, uint32_t& prefixLen, uint32_t& suffixLen) const;
private:
	friend class RegexParser;
	std::vector<Node> m_nodes;
	int m_start;
	bool m_ucp, m_allowEmpty;
	std::map<unsigned,std::pair<int,int> > m_groupFixed;	// group -> (prefix len, suffix len) or (-1,-1)
};

class LexerInstance
{
public:
	LexerInstance() :m_approx(false),m_options(0),m_compiled(false){}
	// PatternLexerInstanceInterface (patternLexer.cpp:971-1141)
	void defineLexem( uint32_t id, const std::string& expression, uint32_t resultIndex, uint32_t level, PosBind posbind);
	void defineSymbol( uint32_t symbolid, uint32_t patternid, const std::string& name);
	uint32_t getSymbol( uint32_t patternid, const std::string& name) const;
	void defineOption( const std::string& name, double value);
	void compile();
	// PatternLexerContextInterface::match (patternLexer.cpp:858-950)
	std::vector<LexemOut> match( const char* src, size_t len) const;
	// the raw report stream the handler is fed with (for tests)
	std::vector<RawMatch> rawMatches( const char* src, size_t len) const;

private:
	struct Def
	{
		std::string expression; uint32_t id, resultIndex, level, editdist; PosBind posbind;
		uint32_t prefixLen, suffixLen;
	};
	struct MatchEvent { uint32_t id; uint8_t level; uint8_t posbind; uint16_t origsize; uint32_t origpos; };
	void handleMatch( std::vector<MatchEvent>& ar, const char* src, uint32_t idx, uint32_t from, uint32_t to) const;
	void handleEvent( std::vector<MatchEvent>& ar, const char* src, uint32_t idx, uint32_t from, uint32_t to) const;
	std::vector<LexemOut> ordinalPositions( const std::vector<MatchEvent>& ar) const;

	// tables with an edit distance pattern (see "approximate literal tables" in l1_oracle.cpp)
	struct ApproxCandidate { uint32_t idx, from, to; };
	bool approxSecondStage( const char* src, size_t len, const ApproxCandidate& c, uint32_t& from, uint32_t& to) const;
	std::vector<LexemOut> matchApprox( const char* src, size_t len) const;
	std::vector<std::vector<uint32_t> > m_literal;	// per definition: its code points (approximate tables only)
	bool m_approx;

	std::vector<Def> m_defs;
	std::vector<Regex> m_regex;
	std::map<uint32_t, std::map<std::string,uint32_t> > m_symbols;	// lexem id -> (name -> symbol id)
	unsigned m_options;
	bool m_compiled;
};

} // namespace
#endif
