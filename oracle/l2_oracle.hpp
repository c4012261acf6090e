// TEST INFRASTRUCTURE ONLY -- CPU oracle for level 2 (token-rule automaton).
//
// This is a CPU restatement of the reference algorithm, written from reading
//   /root/reference/src/ruleMatcherAutomaton.{hpp,cpp}, patternMatcher.cpp and pod*.hpp.
// It is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (struspattern_amd/) never links or calls it.
//
// Why C++ and not plain C: two libstdc++ behaviours are observable in the reference's results
// and are reproduced by using the very same facilities here:
//   * std::unordered_map<uint32_t,uint32_t> iteration order inside ProgramTable::optimize
//     (ruleMatcherAutomaton.cpp:519-532; strus::unordered_map is std::unordered_map in C++11 builds)
//   * std::push_heap / std::pop_heap tie order of the far-expiry queue (cpp:1079-1080, :1100-1127)
//
// Pinning (see DESIGN.md "Oracle"): tests/test_oracle_l2.py checks this restatement against the
// reference's own known-answer test (tests/simpleTokenPatternMatch) and the worked nested example
// recorded in SURVEY.md App. B.10.  The reference L2 sources need strus headers that are absent
// from this image, so the reference itself is not built here (oracle/_ref is not used).
#ifndef SPA_ORACLE_L2_HPP
#define SPA_ORACLE_L2_HPP
#include <stdint.h>
#include <cstddef>
#include <vector>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <stdexcept>

namespace oracle {

typedef uint32_t u32;

// ruleMatcherAutomaton.hpp:48
enum SigType {SigAny=0, SigSequence=1, SigSequenceImm=2, SigWithin=3, SigDel=4, SigAnd=5};
// patternMatcher.cpp:401-440 (order of the switch; numeric values are ours)
enum JoinOp {OpSequence=0, OpSequenceImm=1, OpSequenceStruct=2, OpWithin=3, OpWithinStruct=4, OpAny=5, OpAnd=6};

// patternMatcher.cpp:100-105
enum EventType {TermEvent=0, ExpressionEvent=1, ReferenceEvent=2};
inline u32 eventHandle( EventType t, u32 idx)
{
	if (idx >= (1u<<29)) throw std::runtime_error("event handle out of range");
	return idx | ((u32)t << 29);
}

// ruleMatcherAutomaton.hpp:200-218
struct EventData
{
	u32 start_origseg, end_origseg, start_origpos, end_origpos, start_ordpos, end_ordpos, subdataref, formathandle;
	EventData() :start_origseg(0),end_origseg(0),start_origpos(0),end_origpos(0),start_ordpos(0),end_ordpos(0),subdataref(0),formathandle(0){}
	EventData( u32 sseg, u32 spos, u32 eseg, u32 epos, u32 sord, u32 eord, u32 sub, u32 fmt)
		:start_origseg(sseg),end_origseg(eseg),start_origpos(spos),end_origpos(epos),start_ordpos(sord),end_ordpos(eord),subdataref(sub),formathandle(fmt){}
};

// ruleMatcherAutomaton.hpp:273-289
struct Result
{
	u32 resultHandle, formatHandle, eventDataReferenceIdx, start_ordpos, end_ordpos,
	    start_origseg, end_origseg, start_origpos, end_origpos;
};

struct EventItem { u32 variable; EventData data; };

// --- podStructTableBase.hpp:27-181: growable array + LIFO free list (freed indices are reused
//     last-freed-first).  Index 0 is a valid element index here; callers add +1 where the
//     reference does.
template <class T>
class FreeListTable
{
public:
	FreeListTable() :m_free(0){}
	u32 add( const T& e)
	{
		if (m_free)
		{
			u32 idx = m_free-1;
			m_free = m_next[ idx];
			m_ar[ idx] = e;
			return idx;
		}
		m_ar.push_back( e);
		m_next.push_back( 0);
		return (u32)m_ar.size()-1;
	}
	void remove( u32 idx)
	{
		at( idx);
		m_next[ idx] = m_free;
		m_free = idx+1;
	}
	T& at( u32 idx)
	{
		if (idx >= m_ar.size()) throw std::runtime_error("array bound access (FreeListTable)");
		return m_ar[ idx];
	}
	const T& at( u32 idx) const
	{
		if (idx >= m_ar.size()) throw std::runtime_error("array bound access (FreeListTable)");
		return m_ar[ idx];
	}
	std::size_t size() const {return m_ar.size();}
	void clear() {m_ar.clear(); m_next.clear(); m_free=0;}
private:
	std::vector<T> m_ar;
	std::vector<u32> m_next;
	u32 m_free;
};

// --- podStackPoolBase.hpp:27-178: LIFO singly linked lists living in one FreeListTable.
//     A list handle is (index of head)+1, 0 = empty list.
template <class T>
class ListPool
{
public:
	struct Node { T value; u32 next; };
	void push( u32& lst, const T& v)
	{
		Node n; n.value = v; n.next = lst;
		lst = m_tab.add( n) + 1;
	}
	void removeList( u32 lst)
	{
		while (lst)
		{
			u32 nx = m_tab.at( lst-1).next;
			m_tab.remove( lst-1);
			lst = nx;
		}
	}
	// iteration: returns 0 at end, advances the handle
	const T* nextptr( u32& lst) const
	{
		if (!lst) return 0;
		const Node& n = m_tab.at( lst-1);
		lst = n.next;
		return &n.value;
	}
	bool next( u32& lst, T& v) const
	{
		if (!lst) return false;
		const Node& n = m_tab.at( lst-1);
		v = n.value; lst = n.next;
		return true;
	}
	void clear() {m_tab.clear();}
private:
	FreeListTable<Node> m_tab;
};

// ruleMatcherAutomaton.hpp:291-340
struct ActionSlotDef { u32 initsigval, initcount, event, resultHandle, formatHandle; };
struct TriggerDef { u32 event; unsigned char isKeyEvent; unsigned char sigtype; u32 sigval; u32 variable; };
struct Program { ActionSlotDef slotDef; u32 triggerListIdx; u32 positionRange; };
struct ProgramTrigger { u32 programidx; u32 past_eventid; };

struct OptimizeOptions
{
	float stopwordOccurrenceFactor;
	float weightFactor;
	u32 maxRange;
	OptimizeOptions() :stopwordOccurrenceFactor(0.01f),weightFactor(10.0f),maxRange(5){}
};

// ruleMatcherAutomaton.hpp:344-413
class ProgramTable
{
public:
	ProgramTable() :m_totalNofPrograms(0){}
	void defineEventFrequency( u32 eventid, double df);
	u32 createProgram( u32 positionRange, const ActionSlotDef& def);
	void createTrigger( u32 program, u32 event, bool isKeyEvent, SigType sigtype, u32 sigval, u32 variable);
	void doneProgram( u32 program);
	void defineProgramResult( u32 program, u32 eventid, u32 resultHandle, u32 formatHandle);
	void optimize( const OptimizeOptions& opt);

	const Program& program( u32 programidx) const		{return m_programs.at( programidx-1);}
	const ListPool<TriggerDef>& triggerList() const		{return m_triggerList;}
	u32 getEventProgramList( u32 eventid) const;
	const ProgramTrigger* nextProgramPtr( u32& lst) const	{return m_programTriggerList.nextptr( lst);}
	bool isStopWord( u32 eventid) const			{return m_stopWordSet.find( eventid) != m_stopWordSet.end();}
	std::size_t nofPrograms() const				{return m_programs.size();}
	const std::set<u32>& stopWords() const			{return m_stopWordSet;}
	// introspection for tests: key events in unordered_map iteration order
	std::vector<u32> keyEvents() const;

private:
	void defineEventProgramAlt( u32 eventid, u32 programidx, u32 past_eventid);
	void defineEventProgram( u32 eventid, u32 programidx);
	double calcEventWeight( u32 eventid) const;
	u32 getAltEventId( u32 eventid, u32 triggerListIdx) const;
	void getDelimTokenStopWordSet( u32 triggerListIdx);
	void eliminateUnusedEvents();

	ListPool<TriggerDef> m_triggerList;
	FreeListTable<Program> m_programs;
	ListPool<ProgramTrigger> m_programTriggerList;
	std::unordered_map<u32,u32> m_eventProgramTriggerMap;
	std::set<u32> m_stopWordSet;
	std::map<u32,u32> m_keyOccurrenceMap;
	std::map<u32,u32> m_eventOccurrenceMap;
	std::map<u32,double> m_frequencyMap;
	u32 m_totalNofPrograms;
};

// ruleMatcherAutomaton.hpp:45-74, 87-104, 171-186
struct Trigger { u32 slot; u32 sigtype; u32 variable; u32 sigval; };
struct ActionSlot
{
	u32 value, event, rule, resultHandle, formatHandle, start_ordpos, end_ordpos, start_origseg, start_origpos;
	uint16_t count;
};
struct Rule
{
	u32 actionSlotIdx, eventTriggerListIdx, eventDataReferenceIdx;
	bool done; u32 lastpos;
	bool isActive() const {return actionSlotIdx != 0;}
};
struct EventStruct { EventData data; u32 eventid; };
struct EventLog { EventData data; unsigned int timestmp; };
struct EventDataReference { u32 eventItemListIdx; u32 referenceCount; };
struct DisposeEvent
{
	u32 pos, idx;
	bool operator<( const DisposeEvent& o) const {return pos > o.pos;}
};

// ruleMatcherAutomaton.hpp:132-169 / cpp:34-257
class EventTriggerTable
{
public:
	enum {EventHashTabSize=16, EventHashTabIdxShift=28, EventHashTabIdxMask=15};
	EventTriggerTable() :m_nofTriggers(0){}
	u32 add( u32 event, const Trigger& trigger);
	void remove( u32 idx);
	u32 getTriggerEventId( u32 idx) const;
	const Trigger& getTrigger( u32 idx) const		{return m_triggerTab.at( idx).trigger;}
	void getTriggers( std::vector<u32>& out, u32 event) const;
	u32 nofTriggers() const					{return m_nofTriggers;}
private:
	struct LinkedTrigger { u32 link; Trigger trigger; };
	struct Bucket { std::vector<u32> eventAr; std::vector<u32> ar; };
	Bucket m_bucket[ EventHashTabSize];
	FreeListTable<LinkedTrigger> m_triggerTab;
	u32 m_nofTriggers;
};

// ruleMatcherAutomaton.hpp:431-505 / cpp:589-1334
class StateMachine
{
public:
	explicit StateMachine( const ProgramTable* programTable);

	void doTransition( u32 event, const EventData& data);
	void setCurrentPos( u32 pos);

	const std::vector<Result>& results() const		{return m_results;}
	u32 getEventDataItemListIdx( u32 dataref) const		{return m_eventDataReferenceTable.at( dataref).eventItemListIdx;}
	const EventItem* nextResultItem( u32& lst) const	{return m_eventItemList.nextptr( lst);}

	unsigned int nofProgramsInstalled() const		{return m_nofProgramsInstalled;}
	unsigned int nofAltKeyProgramsInstalled() const		{return m_nofAltKeyProgramsInstalled;}
	unsigned int nofSignalsFired() const			{return m_nofSignalsFired;}
	double nofOpenPatterns() const				{return m_nofOpenPatterns;}

private:
	void fireSignal( u32 slotidx, const Trigger& trigger, const EventData& data,
				std::vector<u32>& disposeRuleList, std::vector<EventStruct>& followList);
	u32 createRule( u32 expiryOrdpos);
	void disposeRule( u32 rule);
	void deactivateRule( u32 rule);
	void disposeEventDataReference( u32 ref);
	void referenceEventData( u32 ref);
	u32 createEventData();
	void appendEventData( u32 ref, const EventItem& item);
	void joinEventData( u32 dest, u32 src);
	void replayPastEvent( u32 eventid, u32 ruleidx, u32 positionRange);
	void installProgram( u32 keyevent, const ProgramTrigger& pt, const EventData& data,
				std::vector<EventStruct>& followList, std::vector<u32>& disposeRuleList);
	void installEventPrograms( u32 keyevent, const EventData& data,
				std::vector<EventStruct>& followList, std::vector<u32>& disposeRuleList);
	void defineDisposeRule( u32 pos, u32 ruleidx);

	enum {DisposeWindowSize=64};
	const ProgramTable* m_programTable;
	EventTriggerTable m_eventTriggerTable;
	FreeListTable<ActionSlot> m_actionSlotTable;
	ListPool<u32> m_eventTriggerList;
	ListPool<EventItem> m_eventItemList;
	FreeListTable<EventDataReference> m_eventDataReferenceTable;
	FreeListTable<Rule> m_ruleTable;
	std::vector<Result> m_results;
	u32 m_curpos;
	u32 m_disposeWindow[ DisposeWindowSize];
	ListPool<u32> m_disposeRuleList;
	std::vector<DisposeEvent> m_ruleDisposeQueue;
	std::map<u32,EventLog> m_stopWordsEventLogMap;
	unsigned int m_nofProgramsInstalled;
	unsigned int m_nofAltKeyProgramsInstalled;
	unsigned int m_nofSignalsFired;
	double m_nofOpenPatterns;
	unsigned int m_timestmp;
};

// Sequential symbol table (strusBase SymbolTable: ids 1,2,3.. in creation order; see
// patternLexer.cpp:285-290 which relies on exactly that).
class SymbolTable
{
public:
	u32 getOrCreate( const std::string& name)
	{
		std::map<std::string,u32>::const_iterator it = m_map.find( name);
		if (it != m_map.end()) return it->second;
		m_keys.push_back( name);
		return m_map[ name] = (u32)m_keys.size();
	}
	u32 get( const std::string& name) const
	{
		std::map<std::string,u32>::const_iterator it = m_map.find( name);
		return it == m_map.end() ? 0 : it->second;
	}
	const std::string& key( u32 id) const {return m_keys.at( id-1);}
	std::size_t size() const {return m_keys.size();}
private:
	std::map<std::string,u32> m_map;
	std::vector<std::string> m_keys;
};

// Input lexem (analyzer::PatternLexem stand-in)
struct Lexem { u32 id, ordpos, origseg, origpos, origsize; };

// `formatHandle` != 0: the reference evaluates that format string over `args` (its subresitemlist,
// patternMatcher.cpp:172-181 / :253-262) into the item / result value; the formatter itself is in
// strusAnalyzer (not in the reference repository), so the oracle keeps handle + arguments.
struct ResultItem
{
	u32 variable, start_ordpos, end_ordpos, start_origseg, start_origpos, end_origseg, end_origpos;
	u32 formatHandle;
	std::vector<ResultItem> args;
};
struct MatchResult
{
	u32 resultHandle, start_ordpos, end_ordpos, start_origseg, start_origpos, end_origseg, end_origpos;
	u32 formatHandle;
	std::vector<ResultItem> items;	// the result's items, or the format arguments when formatHandle != 0 (the reference's item list is empty then)
};

// patternMatcher.cpp:345-733 (PatternMatcherInstance) restated
class MatcherInstance
{
public:
	MatcherInstance() :m_expression_event_cnt(0),m_exclusive(false),m_maxResultSize(100){}

	void defineTermFrequency( u32 termid, double df);
	void pushTerm( u32 termid);
	void pushExpression( JoinOp op, std::size_t argc, u32 range, u32 cardinality);
	void pushPattern( const std::string& name);
	void attachVariable( const std::string& name);
	void definePattern( const std::string& name, const std::string& formatstring, bool visible);
	void defineOption( const std::string& name, double value);
	bool compile();

	const ProgramTable& programTable() const		{return m_programTable;}
	const SymbolTable& patternMap() const			{return m_patternMap;}
	const SymbolTable& variableMap() const			{return m_variableMap;}
	bool exclusive() const					{return m_exclusive;}
	u32 maxResultSize() const				{return m_maxResultSize;}

private:
	struct StackElement
	{
		u32 eventid, program, variable;
		StackElement( u32 e, u32 p=0) :eventid(e),program(p),variable(0){}
	};
	SymbolTable m_variableMap;
	SymbolTable m_patternMap;
	ProgramTable m_programTable;
	std::vector<StackElement> m_stack;
	u32 m_expression_event_cnt;
	OptimizeOptions m_popt;
	bool m_exclusive;
	u32 m_maxResultSize;
	u32 m_nofFormats = 0;
public:
	u32 nofFormats() const {return m_nofFormats;}
private:
};

// patternMatcher.cpp:107-341 (PatternMatcherContext) restated
class MatcherContext
{
public:
	explicit MatcherContext( const MatcherInstance* inst);
	~MatcherContext();
	void putInput( const Lexem& lx);
	std::vector<MatchResult> fetchResults() const;
	void reset();

	unsigned int nofProgramsInstalled() const		{return m_sm->nofProgramsInstalled();}
	unsigned int nofAltKeyProgramsInstalled() const		{return m_sm->nofAltKeyProgramsInstalled();}
	unsigned int nofSignalsFired() const			{return m_sm->nofSignalsFired();}
	double nofOpenPatterns() const				{return m_sm->nofOpenPatterns();}
	unsigned int nofEvents() const				{return m_nofEvents;}

private:
	void gatherResultItems( std::vector<ResultItem>& out, u32 dataref) const;
	std::vector<bool> getCoveredFlags( const std::vector<Result>& results) const;
	const MatcherInstance* m_inst;
	StateMachine* m_sm;
	unsigned int m_nofEvents;
	u32 m_curPosition;
};

} // namespace oracle
#endif
