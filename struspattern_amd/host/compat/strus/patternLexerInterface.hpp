// Stand-ins for strusAnalyzer's PatternLexerInterface / PatternLexerInstanceInterface /
// PatternLexerContextInterface, reconstructed from the overriding declarations in strusPattern
// (src/patternLexer.hpp:30-34, src/patternLexer.cpp:700, :858, :971-1141).
#ifndef _STRUS_ANALYZER_PATTERN_LEXER_INTERFACE_HPP_INCLUDED
#define _STRUS_ANALYZER_PATTERN_LEXER_INTERFACE_HPP_INCLUDED
#include "strus/analyzer/patternLexem.hpp"
#include "strus/structView.hpp"
#include <string>
#include <vector>
namespace strus {
class PatternLexerContextInterface
{
public:
	virtual ~PatternLexerContextInterface(){}
	virtual std::vector<analyzer::PatternLexem> match( const char* src, std::size_t srclen)=0;
	virtual void reset()=0;
};
class PatternLexerInstanceInterface
{
public:
	virtual ~PatternLexerInstanceInterface(){}
	virtual void defineLexemName( unsigned int id, const std::string& name)=0;
	virtual const char* getLexemName( unsigned int id) const=0;
	virtual void defineLexem( unsigned int id, const std::string& expression, unsigned int resultIndex, unsigned int level, analyzer::PositionBind posbind)=0;
	virtual void defineSymbol( unsigned int symbolid, unsigned int patternid, const std::string& name)=0;
	virtual unsigned int getSymbol( unsigned int patternid, const std::string& name) const=0;
	virtual void defineOption( const std::string& name, double value)=0;
	virtual bool compile()=0;
	virtual PatternLexerContextInterface* createContext() const=0;
	virtual const char* name() const=0;
	virtual StructView view() const=0;
};
class PatternLexerInterface
{
public:
	virtual ~PatternLexerInterface(){}
	virtual std::vector<std::string> getCompileOptionNames() const=0;
	virtual PatternLexerInstanceInterface* createInstance() const=0;
	virtual const char* name() const=0;
	virtual StructView view() const=0;
};
}
#endif
