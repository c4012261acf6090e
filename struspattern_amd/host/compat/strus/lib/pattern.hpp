// Exported factory functions of the pattern library: same names and signatures as strusPattern's
// include/strus/lib/pattern.hpp:27-32.
#ifndef _STRUS_PATTERN_LIB_HPP_INCLUDED
#define _STRUS_PATTERN_LIB_HPP_INCLUDED
namespace strus {
class PatternLexerInterface;
class PatternMatcherInterface;
class ErrorBufferInterface;
PatternLexerInterface* createPatternLexer_std( ErrorBufferInterface* errorhnd);
PatternMatcherInterface* createPatternMatcher_std( ErrorBufferInterface* errorhnd);
}
#endif
