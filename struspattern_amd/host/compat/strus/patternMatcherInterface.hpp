// Stand-ins for strusAnalyzer's PatternMatcherInterface / PatternMatcherInstanceInterface /
// PatternMatcherContextInterface, reconstructed from strusPattern (src/patternMatcher.hpp:31-35,
// src/patternMatcher.cpp:131, :271, :303, :320, :361-680).
#ifndef _STRUS_ANALYZER_PATTERN_MATCHER_INTERFACE_HPP_INCLUDED
#define _STRUS_ANALYZER_PATTERN_MATCHER_INTERFACE_HPP_INCLUDED
#include "strus/analyzer/patternLexem.hpp"
#include "strus/analyzer/patternMatcherResult.hpp"
#include "strus/structView.hpp"
#include <string>
#include <vector>
namespace strus {
class PatternMatcherContextInterface
{
public:
	virtual ~PatternMatcherContextInterface(){}
	virtual void putInput( const analyzer::PatternLexem& token)=0;
	virtual std::vector<analyzer::PatternMatcherResult> fetchResults()=0;
	virtual analyzer::PatternMatcherStatistics getStatistics() const=0;
	virtual void reset()=0;
};
class PatternMatcherInstanceInterface
{
public:
	// order of the switch at src/patternMatcher.cpp:401-440 (numeric values unverified, SURVEY.md ch. 4)
	enum JoinOperation {OpSequence, OpSequenceImm, OpSequenceStruct, OpWithin, OpWithinStruct, OpAny, OpAnd};
	virtual ~PatternMatcherInstanceInterface(){}
	virtual void defineTermFrequency( unsigned int termid, double df)=0;
	virtual void pushTerm( unsigned int termid)=0;
	virtual void pushExpression( JoinOperation operation, std::size_t argc, unsigned int range, unsigned int cardinality)=0;
	virtual void pushPattern( const std::string& name)=0;
	virtual void attachVariable( const std::string& name)=0;
	virtual void definePattern( const std::string& name, const std::string& formatstring, bool visible)=0;
	virtual PatternMatcherContextInterface* createContext() const=0;
	virtual void defineOption( const std::string& name, double value)=0;
	virtual bool compile()=0;
	virtual const char* name() const=0;
	virtual StructView view() const=0;
};
class PatternMatcherInterface
{
public:
	virtual ~PatternMatcherInterface(){}
	virtual std::vector<std::string> getCompileOptionNames() const=0;
	virtual PatternMatcherInstanceInterface* createInstance() const=0;
	virtual const char* name() const=0;
	virtual StructView view() const=0;
};
}
#endif
