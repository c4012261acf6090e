// The loadable analyzer module: same file name, same exported object `entryPoint` of type
// strus::AnalyzerModule holding {"std", createPatternLexer_std} and {"std", createPatternMatcher_std}
// as the reference (src/modstrus_analyzer_pattern.cpp:16-24, :59-61).  The strus module loader
// dlopen()s modstrus_analyzer_pattern.so and looks up `entryPoint`.
#include "strus/lib/pattern.hpp"
#include "strus/analyzerModule.hpp"

static const strus::PatternLexerConstructor lexer = { "std", strus::createPatternLexer_std };
static const strus::PatternMatcherConstructor matcher = { "std", strus::createPatternMatcher_std };

static const char* engine_version = "struspattern_amd 0.1 (gfx950): bit-parallel multi-regex lexer + wavefront rule automaton";
static const char* engine_license = " struspattern_amd: no third party regex engine is linked (the reference links Intel Hyperscan here)\n";

extern "C" __attribute__((visibility("default"))) strus::AnalyzerModule entryPoint;
strus::AnalyzerModule entryPoint( lexer, matcher, engine_version, engine_license);
