// Level-2 rule compiler, see l2_compile.hpp.
//
// Behavioural contract (what must come out identical to the reference, cited per function):
//  * the Program / trigger templates produced per operator      src/patternMatcher.cpp:396-505
//  * the order in which programs keyed by one event are visited  (LIFO lists, src/podStackPoolBase.hpp:48-62)
//  * the optimizer's re-keying decisions, which depend on the visiting order of a
//    std::unordered_map<uint32_t,uint32_t>                       src/ruleMatcherAutomaton.cpp:512-586
#include "l2_compile.hpp"
#include "serial.hpp"
#include <algorithm>
#include <cstring>
#include <limits>
#include <stdexcept>

using namespace spa;

namespace {
enum {EV_TERM=0, EV_EXPRESSION=1, EV_REFERENCE=2};
// src/patternMatcher.cpp:100-105
uint32_t eventId( unsigned type, uint32_t idx)
{
	if (idx >= (1u<<29)) throw std::runtime_error( "event handle out of range");
	return idx | ((uint32_t)type << 29);
}
bool sameNoCase( const std::string& a, const char* b)
{
	size_t n = std::strlen( b);
	if (a.size() != n) return false;
	for (size_t i=0; i<n; ++i)
	{
		char x = a[i], y = b[i];
		if (x >= 'A' && x <= 'Z') x += 32;
		if (y >= 'A' && y <= 'Z') y += 32;
		if (x != y) return false;
	}
	return true;
}
enum {OP_SEQUENCE=0, OP_SEQUENCE_IMM=1, OP_SEQUENCE_STRUCT=2, OP_WITHIN=3, OP_WITHIN_STRUCT=4, OP_ANY=5, OP_AND=6};
}

uint32_t SymbolIndex::getOrCreate( const std::string& name)
{
	std::map<std::string,uint32_t>::const_iterator it = m_ids.find( name);
	if (it != m_ids.end()) return it->second;
	m_names.push_back( name);
	return m_ids[ name] = (uint32_t)m_names.size();
}
uint32_t SymbolIndex::get( const std::string& name) const
{
	std::map<std::string,uint32_t>::const_iterator it = m_ids.find( name);
	return it == m_ids.end() ? 0 : it->second;
}
const char* SymbolIndex::key( uint32_t id) const
{
	return (id == 0 || id > m_names.size()) ? 0 : m_names[ id-1].c_str();
}

RuleCompiler::RuleCompiler()
	:m_totalKeyedPrograms(0),m_exprEvents(0),m_formats(0)
	,m_stopwordOccurrenceFactor(0.01f),m_weightFactor(10.0f),m_maxRange(5)	// src/ruleMatcherAutomaton.hpp:373-374
	,m_exclusive(false),m_maxResultSize(100)				// src/patternMatcher.cpp:76-77
{}

// src/patternMatcher.cpp:361-364, src/ruleMatcherAutomaton.cpp:259-266
void RuleCompiler::defineTermFrequency( uint32_t termid, double df)
{
	if (df <= std::numeric_limits<double>::epsilon()) throw std::runtime_error( "illegal value for df (must be positive)");
	m_frequency[ eventId( EV_TERM, termid)] = df;
}

void RuleCompiler::pushTerm( uint32_t termid)
{
	Node n = { eventId( EV_TERM, termid), 0, 0 };
	m_stack.push_back( n);
}

uint32_t RuleCompiler::newProgram( const Prog& p)
{
	m_progs.push_back( p);
	return (uint32_t)m_progs.size();
}

// src/ruleMatcherAutomaton.cpp:303-322: the list of an event grows at its head
void RuleCompiler::addKey( uint32_t event, uint32_t program, uint32_t pastEvent)
{
	KeyRef ref = { program, pastEvent };
	std::unordered_map<uint32_t,uint32_t>::iterator it = m_keymap.find( event);
	if (it == m_keymap.end())
	{
		m_keylists.push_back( std::vector<KeyRef>( 1, ref));
		m_keymap[ event] = (uint32_t)m_keylists.size()-1;
	}
	else
	{
		m_keylists[ it->second].push_back( ref);
	}
	m_keyOccurrence[ event] += 1;
	if (pastEvent)
	{
		m_keyOccurrence[ pastEvent] -= 1;
		m_stopWords.insert( pastEvent);
	}
}

// src/ruleMatcherAutomaton.cpp:280-293: key triggers are visited in installation order
void RuleCompiler::registerKeys( uint32_t program)
{
	const Prog& p = m_progs[ program-1];
	for (size_t ti=p.trigs.size(); ti>0; --ti)
	{
		const Trig& t = p.trigs[ ti-1];
		if (t.isKey)
		{
			addKey( t.event, program, 0);
			++m_totalKeyedPrograms;
		}
	}
}

// Operator -> slot template + trigger templates: table of SURVEY.md App. B.2 = src/patternMatcher.cpp:396-505
void RuleCompiler::pushExpression( int joinop, size_t argc, uint32_t range, uint32_t cardinality)
{
	if (argc > m_stack.size()) throw std::runtime_error( "expression references more arguments than nodes on the stack");
	if (joinop < OP_SEQUENCE || joinop > OP_AND) throw std::runtime_error( "unknown join operation");
	const uint32_t n = (uint32_t)argc;
	Prog p;
	p.initsigval = 0;
	p.initcount = cardinality ? cardinality : n;
	p.event = eventId( EV_EXPRESSION, ++m_exprEvents);
	p.resultHandle = 0; p.formatHandle = 0; p.range = range;
	uint8_t sig = SIG_ANY;
	switch (joinop)
	{
		case OP_SEQUENCE:	sig = SIG_SEQUENCE; p.initsigval = n; break;
		case OP_SEQUENCE_IMM:	sig = SIG_SEQUENCE_IMM; p.initsigval = n; break;
		case OP_SEQUENCE_STRUCT:sig = SIG_SEQUENCE; p.initsigval = n-1; p.initcount -= 1; break;
		case OP_WITHIN:
		case OP_WITHIN_STRUCT:
			if (n > 32) throw std::runtime_error( joinop == OP_WITHIN
				? "operator 'within': number of arguments out of range (32)"
				: "operator 'within_struct': number of arguments out of range (32)");
			sig = SIG_WITHIN; p.initsigval = 0xFFFFFFFFu;
			if (joinop == OP_WITHIN_STRUCT) p.initcount -= 1;
			break;
		case OP_ANY:		sig = SIG_ANY; p.initcount = cardinality ? cardinality : 1; break;
		case OP_AND:		sig = SIG_AND; break;
	}
	const size_t base = m_stack.size() - argc;
	for (uint32_t ai=0; ai<n; ++ai)
	{
		Trig t; t.event = m_stack[ base+ai].event; t.variable = m_stack[ base+ai].variable;
		t.sigtype = sig; t.sigval = 0; t.isKey = false;
		const bool delim = (ai == 0 && (joinop == OP_SEQUENCE_STRUCT || joinop == OP_WITHIN_STRUCT));
		if (delim)
		{
			t.sigtype = SIG_DEL;
		}
		else switch (joinop)
		{
			case OP_SEQUENCE:	t.sigval = n-ai; t.isKey = (ai == 0); break;
			case OP_SEQUENCE_IMM:
				t.sigval = n-ai; t.isKey = (ai == 0);
				if (ai == 0) t.sigtype = SIG_SEQUENCE;
				break;
			case OP_SEQUENCE_STRUCT:t.sigval = n-ai; t.isKey = (ai == 1); break;
			case OP_WITHIN:		t.sigval = 1u << (n-ai-1); t.isKey = true; break;
			case OP_WITHIN_STRUCT:	t.sigval = 1u << (n-ai); t.isKey = true; break;
			default:		t.isKey = true; break;	// any, and
		}
		p.trigs.push_back( t);
	}
	uint32_t program = newProgram( p);
	registerKeys( program);
	m_stack.erase( m_stack.begin()+base, m_stack.end());
	Node node = { p.event, program, 0 };
	m_stack.push_back( node);
}

// src/patternMatcher.cpp:510-520
void RuleCompiler::pushPattern( const std::string& name)
{
	Node n = { eventId( EV_REFERENCE, m_patterns.getOrCreate( name)), 0, 0 };
	m_stack.push_back( n);
}

// src/patternMatcher.cpp:522-543
void RuleCompiler::attachVariable( const std::string& name)
{
	if (m_stack.empty()) throw std::runtime_error( "illegal operation attach variable when no node on the stack");
	if (m_stack.back().variable) throw std::runtime_error( "more than one variable assignment to a node");
	uint32_t id = m_variables.getOrCreate( name);
	if (id >= (1u<<28)) throw std::runtime_error( "too many variables defined");	// src/ruleMatcherAutomaton.hpp:58,69
	m_stack.back().variable = id;
}

// src/patternMatcher.cpp:545-584 (the stack is left as it is)
void RuleCompiler::definePattern( const std::string& name, const std::string& formatstring, bool visible)
{
	if (m_stack.empty()) throw std::runtime_error( "illegal operation close pattern when no node on the stack");
	const Node top = m_stack.back();
	uint32_t handle = m_patterns.getOrCreate( name);
	uint32_t outEvent = eventId( EV_REFERENCE, handle);
	uint32_t format = formatstring.empty() ? 0 : ++m_formats;
	if (format) m_formatStrings.push_back( formatstring);
	uint32_t program = top.program;
	if (!program)
	{
		// a bare term / pattern reference becomes a one-trigger program
		Prog p;
		p.initsigval = 0; p.initcount = 1; p.event = outEvent; p.resultHandle = handle; p.formatHandle = format; p.range = 0;
		Trig t; t.event = top.event; t.isKey = true; t.sigtype = SIG_ANY; t.sigval = 0; t.variable = top.variable;
		p.trigs.push_back( t);
		program = newProgram( p);
		registerKeys( program);
	}
	else if (top.variable)
	{
		throw std::runtime_error( "variable assignments only allowed to subexpressions of pattern");
	}
	Prog& p = m_progs[ program-1];
	p.event = outEvent;
	p.resultHandle = visible ? handle : 0;
	p.formatHandle = format;
}

// src/patternMatcher.cpp:614-644
void RuleCompiler::defineOption( const std::string& name, double value)
{
	const double eps = std::numeric_limits<double>::epsilon();
	if (sameNoCase( name, "stopwordOccurrenceFactor")) m_stopwordOccurrenceFactor = (float)value;
	else if (sameNoCase( name, "weightFactor")) m_weightFactor = (float)value;
	else if (sameNoCase( name, "maxRange")) m_maxRange = (unsigned int)(value + eps);
	else if (sameNoCase( name, "maxResultSize")) m_maxResultSize = (unsigned int)(value + eps);
	else if (sameNoCase( name, "exclusive")) m_exclusive = true;
	else throw std::runtime_error( "unknown token pattern match option: '" + name + "'");
}

// src/ruleMatcherAutomaton.cpp:341-355
double RuleCompiler::eventWeight( uint32_t event) const
{
	double w = 1.0;
	std::map<uint32_t,double>::const_iterator fi = m_frequency.find( event);
	if (fi != m_frequency.end() && fi->second > 0.0) w = fi->second;
	std::map<uint32_t,uint32_t>::const_iterator ki = m_keyOccurrence.find( event);
	if (ki != m_keyOccurrence.end() && ki->second > 0.0) w *= ki->second;
	return w;
}

// src/ruleMatcherAutomaton.cpp:357-430.  Written as a small decision procedure over the triggers
// in installation order; `state` is the signal type of the current candidate (ANY = none yet).
// The And case continues into the sequence/within handling exactly as the reference's missing
// `break` makes it do (:371-391).
uint32_t RuleCompiler::alternativeKey( uint32_t event, const Prog& p) const
{
	uint32_t chosen = 0, chosenVal = 0;
	unsigned state = SIG_ANY;
	for (size_t ti=p.trigs.size(); ti>0; --ti)
	{
		const Trig& t = p.trigs[ ti-1];
		if (t.sigtype == SIG_DEL) continue;
		if (t.sigtype == SIG_ANY) return 0;
		if (t.sigtype == SIG_AND)
		{
			if (state == SIG_AND)
			{
				if (t.event != event) chosen = t.event;
			}
			else if (state == SIG_ANY && t.event != event)
			{
				chosen = t.event; state = SIG_AND;
			}
			else return 0;
		}
		// sequence / sequence_imm / within (and the fall-through from `and`)
		const bool other = (t.event != event);
		if (state == t.sigtype || (state == SIG_SEQUENCE_IMM && t.sigtype == SIG_SEQUENCE))
		{
			if (chosenVal < t.sigval && other) { chosen = t.event; chosenVal = t.sigval; state = t.sigtype; }
		}
		else if (state == SIG_ANY && other)
		{
			chosen = t.event; chosenVal = t.sigval; state = t.sigtype;
		}
		else return 0;
	}
	return chosen;
}

// src/ruleMatcherAutomaton.cpp:478-510
void RuleCompiler::dropUnlistenedEvents()
{
	std::set<uint32_t> listened;
	std::set<uint32_t> reachable;
	for (std::unordered_map<uint32_t,uint32_t>::const_iterator ki=m_keymap.begin(); ki!=m_keymap.end(); ++ki)
	{
		const std::vector<KeyRef>& lst = m_keylists[ ki->second];
		for (size_t i=0; i<lst.size(); ++i)
		{
			reachable.insert( lst[i].program);
			const Prog& p = m_progs[ lst[i].program-1];
			for (size_t ti=0; ti<p.trigs.size(); ++ti) listened.insert( p.trigs[ti].event);
		}
	}
	for (std::set<uint32_t>::const_iterator pi=reachable.begin(); pi!=reachable.end(); ++pi)
	{
		Prog& p = m_progs[ *pi-1];
		if (!listened.count( p.event)) p.event = 0;
	}
}

// src/patternMatcher.cpp:646-671 -> src/ruleMatcherAutomaton.cpp:512-586
void RuleCompiler::compile()
{
	dropUnlistenedEvents();

	std::vector<uint32_t> candidates;
	for (std::unordered_map<uint32_t,uint32_t>::const_iterator ki=m_keymap.begin(); ki!=m_keymap.end(); ++ki)
	{
		std::map<uint32_t,uint32_t>::const_iterator oi = m_keyOccurrence.find( ki->first);
		if (oi != m_keyOccurrence.end() && oi->second >= (float)m_totalKeyedPrograms * m_stopwordOccurrenceFactor)
		{
			candidates.push_back( ki->first);
		}
	}
	for (size_t ci=0; ci<candidates.size(); ++ci)
	{
		const uint32_t event = candidates[ ci];
		std::unordered_map<uint32_t,uint32_t>::iterator ki = m_keymap.find( event);
		if (ki == m_keymap.end()) continue;
		const uint32_t listIdx = ki->second;
		const std::vector<KeyRef> visiting( m_keylists[ listIdx].rbegin(), m_keylists[ listIdx].rend());
		std::vector<KeyRef> kept;
		const double weight = eventWeight( event);
		for (size_t i=0; i<visiting.size(); ++i)
		{
			const KeyRef ref = visiting[ i];
			const Prog& p = m_progs[ ref.program-1];
			const uint32_t alt = alternativeKey( event, p);
			bool moved = false;
			if (alt)
			{
				double altWeight = eventWeight( alt) * m_weightFactor;
				if (!ref.pastEvent && m_maxRange >= p.range && weight > altWeight)
				{
					addKey( alt, ref.program, event);
					for (size_t ti=0; ti<p.trigs.size(); ++ti)
					{
						if (p.trigs[ti].sigtype == SIG_DEL) m_stopWords.insert( p.trigs[ti].event);
					}
					moved = true;
				}
			}
			if (!moved) kept.push_back( ref);
		}
		// addKey may have grown m_keylists / rehashed m_keymap: look the entry up again
		ki = m_keymap.find( event);
		if (kept.empty())
		{
			m_keymap.erase( ki);
			m_keylists[ listIdx].clear();
		}
		else
		{
			m_keylists[ listIdx] = kept;
		}
	}
}

void RuleCompiler::flatten( FlatTables& out) const
{
	out.programs.clear(); out.trigdefs.clear(); out.keytab.clear(); out.keylist.clear();
	out.maxTrigCount = 0;
	std::map<uint32_t,uint32_t> stopIdx;
	for (std::set<uint32_t>::const_iterator si=m_stopWords.begin(); si!=m_stopWords.end(); ++si)
	{
		uint32_t idx = (uint32_t)stopIdx.size()+1;
		stopIdx[ *si] = idx;
	}
	out.nofStopWords = (uint32_t)stopIdx.size();
	for (size_t pi=0; pi<m_progs.size(); ++pi)
	{
		const Prog& p = m_progs[ pi];
		DevProgram d;
		d.initsigval = p.initsigval; d.initcount = p.initcount; d.event = p.event; d.resultHandle = p.resultHandle;
		d.formatHandle = p.formatHandle; d.positionRange = p.range;
		d.trigBegin = (uint32_t)out.trigdefs.size(); d.trigCount = (uint32_t)p.trigs.size();
		if (d.trigCount > out.maxTrigCount) out.maxTrigCount = d.trigCount;
		for (size_t ti=p.trigs.size(); ti>0; --ti)
		{
			const Trig& t = p.trigs[ ti-1];
			DevTrigDef td; td.event = t.event; td.sigval = t.sigval; td.variable = t.variable;
			td.flags = (uint32_t)t.sigtype | (t.isKey ? 0x100u : 0u);
			out.trigdefs.push_back( td);
		}
		out.programs.push_back( d);
	}
	// hash table over key events and stop words
	std::set<uint32_t> events;
	for (std::unordered_map<uint32_t,uint32_t>::const_iterator ki=m_keymap.begin(); ki!=m_keymap.end(); ++ki) events.insert( ki->first);
	for (std::map<uint32_t,uint32_t>::const_iterator si=stopIdx.begin(); si!=stopIdx.end(); ++si) events.insert( si->first);
	size_t tabsize = 16;
	while (tabsize < events.size()*2+2) tabsize <<= 1;
	DevKeyEntry empty = {0,0,0,0};
	out.keytab.assign( tabsize, empty);
	for (std::set<uint32_t>::const_iterator ei=events.begin(); ei!=events.end(); ++ei)
	{
		DevKeyEntry e; e.event = *ei; e.listBegin = (uint32_t)out.keylist.size(); e.listCount = 0; e.stopIdx = 0;
		std::unordered_map<uint32_t,uint32_t>::const_iterator ki = m_keymap.find( *ei);
		if (ki != m_keymap.end())
		{
			const std::vector<KeyRef>& lst = m_keylists[ ki->second];
			for (size_t i=lst.size(); i>0; --i)
			{
				DevKeyRef r; r.program = lst[i-1].program-1; r.pastEvent = lst[i-1].pastEvent; r.pastStopIdx = 0; r._pad = 0;
				if (r.pastEvent)
				{
					std::map<uint32_t,uint32_t>::const_iterator si = stopIdx.find( r.pastEvent);
					r.pastStopIdx = si == stopIdx.end() ? 0 : si->second;
				}
				out.keylist.push_back( r);
			}
			e.listCount = (uint32_t)lst.size();
		}
		std::map<uint32_t,uint32_t>::const_iterator si = stopIdx.find( *ei);
		if (si != stopIdx.end()) e.stopIdx = si->second;
		size_t slot = keyHash( *ei) & (tabsize-1);
		while (out.keytab[ slot].event) slot = (slot+1) & (tabsize-1);
		out.keytab[ slot] = e;
	}
	if (out.keylist.empty()) { DevKeyRef r = {0,0,0,0}; out.keylist.push_back( r); }
	if (out.trigdefs.empty()) { DevTrigDef t = {0,0,0,0}; out.trigdefs.push_back( t); }
	if (out.programs.empty()) { DevProgram p; std::memset( &p, 0, sizeof(p)); out.programs.push_back( p); }
}

std::vector<uint32_t> RuleCompiler::dump() const
{
	std::vector<uint32_t> buf;
	std::vector<uint32_t> keys;
	for (std::unordered_map<uint32_t,uint32_t>::const_iterator ki=m_keymap.begin(); ki!=m_keymap.end(); ++ki) keys.push_back( ki->first);
	std::sort( keys.begin(), keys.end());
	buf.push_back( (uint32_t)m_progs.size());
	buf.push_back( (uint32_t)keys.size());
	buf.push_back( (uint32_t)m_stopWords.size());
	for (size_t pi=0; pi<m_progs.size(); ++pi)
	{
		const Prog& p = m_progs[ pi];
		buf.push_back( p.initsigval); buf.push_back( p.initcount); buf.push_back( p.event);
		buf.push_back( p.resultHandle); buf.push_back( p.formatHandle); buf.push_back( p.range);
		buf.push_back( (uint32_t)p.trigs.size());
		for (size_t ti=p.trigs.size(); ti>0; --ti)
		{
			const Trig& t = p.trigs[ ti-1];
			buf.push_back( t.event); buf.push_back( t.isKey ? 1 : 0); buf.push_back( t.sigtype);
			buf.push_back( t.sigval); buf.push_back( t.variable);
		}
	}
	for (size_t ki=0; ki<keys.size(); ++ki)
	{
		const std::vector<KeyRef>& lst = m_keylists[ m_keymap.find( keys[ki])->second];
		buf.push_back( keys[ki]);
		buf.push_back( (uint32_t)lst.size());
		for (size_t i=lst.size(); i>0; --i) { buf.push_back( lst[i-1].program); buf.push_back( lst[i-1].pastEvent); }
	}
	for (std::set<uint32_t>::const_iterator si=m_stopWords.begin(); si!=m_stopWords.end(); ++si) buf.push_back( *si);
	return buf;
}

// ---------------------------------------------------------------- the rule set as a blob
static const char L2_MAGIC[ 9] = "SPAL2v01";

void RuleCompiler::save( std::vector<uint8_t>& out, bool compiled) const
{
	BlobWriter w( L2_MAGIC);
	w.u32( compiled ? 1u : 0u);
	w.f32( m_stopwordOccurrenceFactor); w.f32( m_weightFactor); w.u32( m_maxRange); w.u32( m_exclusive ? 1u : 0u); w.u32( m_maxResultSize);
	w.u32( m_exprEvents); w.u32( m_totalKeyedPrograms);
	w.u32( (uint32_t)m_patterns.names().size());
	for (size_t i=0; i<m_patterns.names().size(); ++i) w.str( m_patterns.names()[ i]);
	w.u32( (uint32_t)m_variables.names().size());
	for (size_t i=0; i<m_variables.names().size(); ++i) w.str( m_variables.names()[ i]);
	w.u32( (uint32_t)m_formatStrings.size());
	for (size_t i=0; i<m_formatStrings.size(); ++i) w.str( m_formatStrings[ i]);
	w.u32( (uint32_t)m_progs.size());
	for (size_t pi=0; pi<m_progs.size(); ++pi)
	{
		const Prog& p = m_progs[ pi];
		w.u32( p.initsigval); w.u32( p.initcount); w.u32( p.event); w.u32( p.resultHandle); w.u32( p.formatHandle); w.u32( p.range);
		w.u32( (uint32_t)p.trigs.size());
		for (size_t ti=0; ti<p.trigs.size(); ++ti)
		{
			const Trig& t = p.trigs[ ti];
			w.u32( t.event); w.u32( t.isKey ? 1u : 0u); w.u32( t.sigtype); w.u32( t.sigval); w.u32( t.variable);
		}
	}
	// key index: lists in push order; the events in ascending order (the map's own order only matters to the optimizer)
	std::vector<uint32_t> keys;
	for (std::unordered_map<uint32_t,uint32_t>::const_iterator ki=m_keymap.begin(); ki!=m_keymap.end(); ++ki) keys.push_back( ki->first);
	std::sort( keys.begin(), keys.end());
	w.u32( (uint32_t)keys.size());
	for (size_t k=0; k<keys.size(); ++k)
	{
		const std::vector<KeyRef>& lst = m_keylists[ m_keymap.find( keys[ k])->second];
		w.u32( keys[ k]); w.u32( (uint32_t)lst.size());
		for (size_t i=0; i<lst.size(); ++i) { w.u32( lst[ i].program); w.u32( lst[ i].pastEvent); }
	}
	w.u32( (uint32_t)m_stopWords.size());
	for (std::set<uint32_t>::const_iterator si=m_stopWords.begin(); si!=m_stopWords.end(); ++si) w.u32( *si);
	w.u32( (uint32_t)m_keyOccurrence.size());
	for (std::map<uint32_t,uint32_t>::const_iterator oi=m_keyOccurrence.begin(); oi!=m_keyOccurrence.end(); ++oi) { w.u32( oi->first); w.u32( oi->second); }
	w.u32( (uint32_t)m_frequency.size());
	for (std::map<uint32_t,double>::const_iterator fi=m_frequency.begin(); fi!=m_frequency.end(); ++fi) { w.u32( fi->first); w.f64( fi->second); }
	out.swap( w.finish());
}

bool RuleCompiler::load( const void* blob, size_t size)
{
	BlobReader r( blob, size, L2_MAGIC);
	RuleCompiler n;
	const bool compiled = r.u32() != 0;
	n.m_stopwordOccurrenceFactor = r.f32(); n.m_weightFactor = r.f32(); n.m_maxRange = r.u32(); n.m_exclusive = r.u32() != 0; n.m_maxResultSize = r.u32();
	n.m_exprEvents = r.u32(); n.m_totalKeyedPrograms = r.u32();
	const uint32_t np = r.u32();
	for (uint32_t i=0; i<np; ++i) n.m_patterns.getOrCreate( r.str());
	const uint32_t nv = r.u32();
	for (uint32_t i=0; i<nv; ++i) n.m_variables.getOrCreate( r.str());
	const uint32_t nf = r.u32();
	for (uint32_t i=0; i<nf; ++i) n.m_formatStrings.push_back( r.str());
	n.m_formats = nf;
	const uint32_t ng = r.u32();
	for (uint32_t pi=0; pi<ng; ++pi)
	{
		Prog p;
		p.initsigval = r.u32(); p.initcount = r.u32(); p.event = r.u32(); p.resultHandle = r.u32(); p.formatHandle = r.u32(); p.range = r.u32();
		const uint32_t nt = r.u32();
		for (uint32_t ti=0; ti<nt; ++ti)
		{
			Trig t; t.event = r.u32(); t.isKey = r.u32() != 0; t.sigtype = (uint8_t)r.u32(); t.sigval = r.u32(); t.variable = r.u32();
			if (t.sigtype > SIG_AND || t.variable > nv) throw std::runtime_error( "rule set blob holds an invalid trigger");
			p.trigs.push_back( t);
		}
		if (p.resultHandle > np || p.formatHandle > nf) throw std::runtime_error( "rule set blob holds an invalid program");
		n.m_progs.push_back( p);
	}
	const uint32_t nk = r.u32();
	for (uint32_t k=0; k<nk; ++k)
	{
		const uint32_t ev = r.u32(), cnt = r.u32();
		n.m_keymap[ ev] = (uint32_t)n.m_keylists.size();
		n.m_keylists.push_back( std::vector<KeyRef>());
		for (uint32_t i=0; i<cnt; ++i)
		{
			KeyRef kr; kr.program = r.u32(); kr.pastEvent = r.u32();
			if (kr.program == 0 || kr.program > ng) throw std::runtime_error( "rule set blob holds an invalid key list");
			n.m_keylists.back().push_back( kr);
		}
	}
	const uint32_t ns = r.u32();
	for (uint32_t i=0; i<ns; ++i) n.m_stopWords.insert( r.u32());
	const uint32_t no = r.u32();
	for (uint32_t i=0; i<no; ++i) { const uint32_t ev = r.u32(); n.m_keyOccurrence[ ev] = r.u32(); }
	const uint32_t nq = r.u32();
	for (uint32_t i=0; i<nq; ++i) { const uint32_t ev = r.u32(); n.m_frequency[ ev] = r.f64(); }
	if (!r.atEnd()) throw std::runtime_error( "rule set blob has trailing data");
	*this = n;
	return compiled;
}
