// Stand-in for strusModule's strus/analyzerModule.hpp: the object the module loader looks up as
// `entryPoint` (strusPattern src/modstrus_analyzer_pattern.cpp:16-24, :59-61).
#ifndef _STRUS_MODULE_ANALYZER_HPP_INCLUDED
#define _STRUS_MODULE_ANALYZER_HPP_INCLUDED
namespace strus {
class ErrorBufferInterface;
class PatternLexerInterface;
class PatternMatcherInterface;
struct PatternLexerConstructor
{
	typedef PatternLexerInterface* (*Create)( ErrorBufferInterface* errorhnd);
	const char* name;
	Create create;
};
struct PatternMatcherConstructor
{
	typedef PatternMatcherInterface* (*Create)( ErrorBufferInterface* errorhnd);
	const char* name;
	Create create;
};
struct AnalyzerModule
{
	AnalyzerModule( const PatternLexerConstructor& lexer_, const PatternMatcherConstructor& matcher_, const char* version_3rdparty, const char* license_3rdparty)
		:patternLexer(&lexer_),patternMatcher(&matcher_),version3rdparty(version_3rdparty),license3rdparty(license_3rdparty){}
	const PatternLexerConstructor* patternLexer;
	const PatternMatcherConstructor* patternMatcher;
	const char* version3rdparty;
	const char* license3rdparty;
};
}
#endif
