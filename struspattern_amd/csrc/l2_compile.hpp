// Level-2 rule compiler (host side of the product): expression-stack API -> flat, GPU-resident
// ProgramTable.  Replaces PatternMatcherInstance + ProgramTable of the reference
// (src/patternMatcher.cpp:345-733, src/ruleMatcherAutomaton.cpp:259-586) with a design aimed at
// the device: rules are kept as plain vectors on the host and flattened into CSR arrays plus one
// open-addressing hash table that the HIP kernel reads straight from HBM/L2.
#ifndef SPA_L2_COMPILE_HPP
#define SPA_L2_COMPILE_HPP
#include <stdint.h>
#include <cstddef>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>
#include "l2_tables.h"

namespace spa {

struct FlatTables
{
	std::vector<DevProgram> programs;
	std::vector<DevTrigDef> trigdefs;
	std::vector<DevKeyEntry> keytab;	// size is a power of two
	std::vector<DevKeyRef> keylist;
	uint32_t nofStopWords;
	uint32_t maxTrigCount;
};

class SymbolIndex
{
public:
	uint32_t getOrCreate( const std::string& name);
	uint32_t get( const std::string& name) const;
	const char* key( uint32_t id) const;
	const std::vector<std::string>& names() const {return m_names;}
private:
	std::map<std::string,uint32_t> m_ids;
	std::vector<std::string> m_names;
};

class RuleCompiler
{
public:
	RuleCompiler();

	// PatternMatcherInstanceInterface (src/patternMatcher.cpp:361-671); throw std::runtime_error
	void defineTermFrequency( uint32_t termid, double df);
	void pushTerm( uint32_t termid);
	void pushExpression( int joinop, size_t argc, uint32_t range, uint32_t cardinality);
	void pushPattern( const std::string& name);
	void attachVariable( const std::string& name);
	void definePattern( const std::string& name, const std::string& formatstring, bool visible);
	void defineOption( const std::string& name, double value);
	void compile();			// the optimizer of src/ruleMatcherAutomaton.cpp:512-586

	bool exclusive() const			{return m_exclusive;}
	uint32_t maxResultSize() const		{return m_maxResultSize;}
	const SymbolIndex& patterns() const	{return m_patterns;}
	const SymbolIndex& variables() const	{return m_variables;}
	uint32_t formatCount() const		{return m_formats;}
	const char* formatString( uint32_t handle) const	{return (handle && handle <= m_formatStrings.size()) ? m_formatStrings[ handle-1].c_str() : 0;}

	void flatten( FlatTables& out) const;
	// the rule set as a blob and back (serial.hpp): tables, names, format strings, options
	void save( std::vector<uint8_t>& out, bool compiled) const;
	bool load( const void* blob, size_t size);		// returns the `compiled` flag of save()
	// canonical dump (same format as oracle's orc_l2_dump_table, see oracle/oracle_capi.cpp)
	std::vector<uint32_t> dump() const;

private:
	struct Trig { uint32_t event; bool isKey; uint8_t sigtype; uint32_t sigval; uint32_t variable; };
	struct Prog
	{
		uint32_t initsigval, initcount, event, resultHandle, formatHandle, range;
		std::vector<Trig> trigs;	// creation order; installation order is the reverse
	};
	struct KeyRef { uint32_t program; uint32_t pastEvent; };	// program is 1-based as in the reference
	struct Node { uint32_t event; uint32_t program; uint32_t variable; };

	uint32_t newProgram( const Prog& p);
	void registerKeys( uint32_t program);
	void addKey( uint32_t event, uint32_t program, uint32_t pastEvent);
	double eventWeight( uint32_t event) const;
	uint32_t alternativeKey( uint32_t event, const Prog& p) const;
	void dropUnlistenedEvents();

	SymbolIndex m_patterns;
	SymbolIndex m_variables;
	std::vector<Prog> m_progs;
	// key index: the unordered_map only fixes the visiting order of optimize(); the mapped value
	// is an index into m_keylists (push order; iteration order is the reverse)
	std::unordered_map<uint32_t,uint32_t> m_keymap;
	std::vector<std::vector<KeyRef> > m_keylists;
	std::set<uint32_t> m_stopWords;
	std::map<uint32_t,uint32_t> m_keyOccurrence;
	std::map<uint32_t,double> m_frequency;
	uint32_t m_totalKeyedPrograms;
	std::vector<Node> m_stack;
	uint32_t m_exprEvents;
	uint32_t m_formats;
	std::vector<std::string> m_formatStrings;
	float m_stopwordOccurrenceFactor;
	float m_weightFactor;
	uint32_t m_maxRange;
	bool m_exclusive;
	uint32_t m_maxResultSize;
};

} // namespace
#endif
