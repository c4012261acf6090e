// TEST INFRASTRUCTURE ONLY -- CPU oracle for level 2; see l2_oracle.hpp for scope and pinning.
// Every function cites the reference lines it restates (paths relative to /root/reference/src).
#include "l2_oracle.hpp"
#include <algorithm>
#include <limits>
#include <cstring>

using namespace oracle;

// ------------------------------------------------------------------ EventTriggerTable
// ruleMatcherAutomaton.cpp:34-40
static inline u32 evhash( u32 a)
{
	a += ~(a>>5);
	a +=  (a<<3);
	a ^=  (a>>4);
	return a;
}

// ruleMatcherAutomaton.cpp:114-131
u32 EventTriggerTable::add( u32 event, const Trigger& trigger)
{
	u32 htidx = evhash( event) & EventHashTabIdxMask;
	Bucket& rec = m_bucket[ htidx];
	u32 pos = (u32)rec.eventAr.size();
	if (pos >= (1u<<EventHashTabIdxShift)) throw std::runtime_error("too many event triggers defined");
	LinkedTrigger lt; lt.link = pos + (htidx << EventHashTabIdxShift); lt.trigger = trigger;
	u32 rt = m_triggerTab.add( lt);
	rec.eventAr.push_back( event);
	rec.ar.push_back( rt);
	++m_nofTriggers;
	return rt;
}

// ruleMatcherAutomaton.cpp:133-152 (swap-with-last removal)
void EventTriggerTable::remove( u32 idx)
{
	u32 link = m_triggerTab.at( idx).link;
	u32 htidx = (link >> EventHashTabIdxShift) & EventHashTabIdxMask;
	u32 aridx = link & ((1u << EventHashTabIdxShift) -1);
	Bucket& rec = m_bucket[ htidx];
	if (aridx >= rec.ar.size() || rec.ar[ aridx] != idx)
	{
		throw std::runtime_error("bad trigger index (remove trigger)");
	}
	m_triggerTab.remove( idx);
	u32 last = (u32)rec.ar.size()-1;
	if (aridx != last)
	{
		rec.eventAr[ aridx] = rec.eventAr[ last];
		rec.ar[ aridx] = rec.ar[ last];
		m_triggerTab.at( rec.ar[ aridx]).link = link;
	}
	rec.eventAr.pop_back();
	rec.ar.pop_back();
	--m_nofTriggers;
}

// ruleMatcherAutomaton.cpp:154-161
u32 EventTriggerTable::getTriggerEventId( u32 idx) const
{
	u32 link = m_triggerTab.at( idx).link;
	u32 htidx = (link >> EventHashTabIdxShift) & EventHashTabIdxMask;
	u32 aridx = link & ((1u << EventHashTabIdxShift) -1);
	return m_bucket[ htidx].eventAr.at( aridx);
}

// ruleMatcherAutomaton.cpp:229-257 (the SSE and the scalar scan visit matches in ascending array order)
void EventTriggerTable::getTriggers( std::vector<u32>& out, u32 event) const
{
	if (!event) return;
	const Bucket& rec = m_bucket[ evhash( event) & EventHashTabIdxMask];
	for (std::size_t wi=0; wi<rec.eventAr.size(); ++wi)
	{
		if (rec.eventAr[ wi] == event) out.push_back( rec.ar[ wi]);
	}
}

// ------------------------------------------------------------------ ProgramTable
// ruleMatcherAutomaton.cpp:259-266
void ProgramTable::defineEventFrequency( u32 eventid, double df)
{
	if (df <= std::numeric_limits<double>::epsilon()) throw std::runtime_error("illegal value for df (must be positive)");
	m_frequencyMap[ eventid] = df;
}

// ruleMatcherAutomaton.cpp:268-271
u32 ProgramTable::createProgram( u32 positionRange, const ActionSlotDef& def)
{
	Program p; p.slotDef = def; p.triggerListIdx = 0; p.positionRange = positionRange;
	return 1 + m_programs.add( p);
}

// ruleMatcherAutomaton.cpp:273-278
void ProgramTable::createTrigger( u32 programidx, u32 event, bool isKeyEvent, SigType sigtype, u32 sigval, u32 variable)
{
	Program& prg = m_programs.at( programidx-1);
	TriggerDef td; td.event = event; td.isKeyEvent = (unsigned char)isKeyEvent; td.sigtype = (unsigned char)sigtype;
	td.sigval = sigval; td.variable = variable;
	m_triggerList.push( prg.triggerListIdx, td);
	m_eventOccurrenceMap[ event] += 1;
}

// ruleMatcherAutomaton.cpp:280-293
void ProgramTable::doneProgram( u32 programidx)
{
	u32 lst = m_programs.at( programidx-1).triggerListIdx;
	const TriggerDef* td;
	while (0!=(td=m_triggerList.nextptr( lst)))
	{
		if (td->isKeyEvent) defineEventProgram( td->event, programidx);
	}
}

// ruleMatcherAutomaton.cpp:295-301
void ProgramTable::defineProgramResult( u32 programidx, u32 eventid, u32 resultHandle, u32 formatHandle)
{
	Program& prg = m_programs.at( programidx-1);
	prg.slotDef.event = eventid;
	prg.slotDef.resultHandle = resultHandle;
	prg.slotDef.formatHandle = formatHandle;
}

// ruleMatcherAutomaton.cpp:303-322
void ProgramTable::defineEventProgramAlt( u32 eventid, u32 programidx, u32 past_eventid)
{
	ProgramTrigger pt; pt.programidx = programidx; pt.past_eventid = past_eventid;
	std::unordered_map<u32,u32>::iterator ei = m_eventProgramTriggerMap.find( eventid);
	if (ei == m_eventProgramTriggerMap.end())
	{
		u32 prglist = 0;
		m_programTriggerList.push( prglist, pt);
		m_eventProgramTriggerMap[ eventid] = prglist;
	}
	else
	{
		m_programTriggerList.push( ei->second, pt);
	}
	m_keyOccurrenceMap[ eventid] += 1;
	if (past_eventid)
	{
		m_keyOccurrenceMap[ past_eventid] -= 1;
		m_stopWordSet.insert( past_eventid);
	}
}

// ruleMatcherAutomaton.cpp:324-328
void ProgramTable::defineEventProgram( u32 eventid, u32 programidx)
{
	defineEventProgramAlt( eventid, programidx, 0);
	++m_totalNofPrograms;
}

// ruleMatcherAutomaton.cpp:330-334
u32 ProgramTable::getEventProgramList( u32 eventid) const
{
	std::unordered_map<u32,u32>::const_iterator ei = m_eventProgramTriggerMap.find( eventid);
	return ei == m_eventProgramTriggerMap.end() ? 0 : ei->second;
}

std::vector<u32> ProgramTable::keyEvents() const
{
	std::vector<u32> rt;
	for (std::unordered_map<u32,u32>::const_iterator ei = m_eventProgramTriggerMap.begin(); ei != m_eventProgramTriggerMap.end(); ++ei) rt.push_back( ei->first);
	return rt;
}

// ruleMatcherAutomaton.cpp:341-355 (note: keyOccurrence is unsigned; "> 0.0" is as written)
double ProgramTable::calcEventWeight( u32 eventid) const
{
	double kf = 1.0;
	std::map<u32,double>::const_iterator fi = m_frequencyMap.find( eventid);
	if (fi != m_frequencyMap.end() && fi->second > 0.0) kf = fi->second;
	std::map<u32,u32>::const_iterator ki = m_keyOccurrenceMap.find( eventid);
	if (ki != m_keyOccurrenceMap.end() && ki->second > 0.0) kf *= ki->second;
	return kf;
}

// ruleMatcherAutomaton.cpp:357-430 -- including the fall-through of `case SigAnd` into the
// sequence/within case (no break at :390).
u32 ProgramTable::getAltEventId( u32 eventid, u32 lst) const
{
	const TriggerDef* td;
	u32 eventid_selected = 0;
	u32 sigval_selected = 0;
	SigType sigtype = SigAny;

	while (0!=(td=m_triggerList.nextptr( lst)))
	{
		switch ((SigType)td->sigtype)
		{
			case SigAny:
				return 0;
			case SigAnd:
				if (sigtype == (SigType)td->sigtype)
				{
					if (td->event != eventid)
					{
						eventid_selected = td->event;
						sigtype = (SigType)td->sigtype;
					}
				}
				else if (sigtype == SigAny && td->event != eventid)
				{
					eventid_selected = td->event;
					sigtype = (SigType)td->sigtype;
				}
				else
				{
					return 0;
				}
				/* fall through (as in the reference) */
			case SigSequence:
			case SigSequenceImm:
			case SigWithin:
				if (sigtype == (SigType)td->sigtype)
				{
					if (sigval_selected < td->sigval && td->event != eventid)
					{
						eventid_selected = td->event;
						sigval_selected = td->sigval;
						sigtype = (SigType)td->sigtype;
					}
				}
				else if (sigtype == SigAny && td->event != eventid)
				{
					eventid_selected = td->event;
					sigval_selected = td->sigval;
					sigtype = (SigType)td->sigtype;
				}
				else if (sigtype == SigSequenceImm && td->sigtype == SigSequence)
				{
					if (sigval_selected < td->sigval && td->event != eventid)
					{
						eventid_selected = td->event;
						sigval_selected = td->sigval;
						sigtype = (SigType)td->sigtype;
					}
				}
				else
				{
					return 0;
				}
				break;
			case SigDel:
				break;
		}
	}
	return eventid_selected;
}

// ruleMatcherAutomaton.cpp:432-442
void ProgramTable::getDelimTokenStopWordSet( u32 lst)
{
	const TriggerDef* td;
	while (0!=(td=m_triggerList.nextptr( lst)))
	{
		if (td->sigtype == SigDel) m_stopWordSet.insert( td->event);
	}
}

// ruleMatcherAutomaton.cpp:478-510
void ProgramTable::eliminateUnusedEvents()
{
	std::set<u32> usedEvents;
	std::set<u32> programs;
	for (std::unordered_map<u32,u32>::const_iterator ei = m_eventProgramTriggerMap.begin(); ei != m_eventProgramTriggerMap.end(); ++ei)
	{
		u32 prglist = ei->second;
		const ProgramTrigger* pt;
		while (0!=(pt=m_programTriggerList.nextptr( prglist)))
		{
			programs.insert( pt->programidx);
			u32 tl = m_programs.at( pt->programidx-1).triggerListIdx;
			const TriggerDef* td;
			while (0!=(td=m_triggerList.nextptr( tl))) usedEvents.insert( td->event);
		}
	}
	for (std::set<u32>::const_iterator gi = programs.begin(); gi != programs.end(); ++gi)
	{
		Program& prg = m_programs.at( *gi-1);
		if (usedEvents.find( prg.slotDef.event) == usedEvents.end()) prg.slotDef.event = 0;
	}
}

// ruleMatcherAutomaton.cpp:512-586
void ProgramTable::optimize( const OptimizeOptions& opt)
{
	eliminateUnusedEvents();

	std::vector<u32> eventsToMove;
	for (std::unordered_map<u32,u32>::const_iterator ei = m_eventProgramTriggerMap.begin(); ei != m_eventProgramTriggerMap.end(); ++ei)
	{
		u32 eventid = ei->first;
		std::map<u32,u32>::const_iterator ki = m_keyOccurrenceMap.find( eventid);
		if (ki != m_keyOccurrenceMap.end()
		&&  ki->second >= (float)m_totalNofPrograms * opt.stopwordOccurrenceFactor)
		{
			eventsToMove.push_back( eventid);
		}
	}
	for (std::vector<u32>::const_iterator mi = eventsToMove.begin(); mi != eventsToMove.end(); ++mi)
	{
		std::unordered_map<u32,u32>::iterator ei = m_eventProgramTriggerMap.find( *mi);
		if (ei == m_eventProgramTriggerMap.end()) continue;
		u32 eventid = ei->first;
		u32 prglist = ei->second;
		u32 new_prglist = 0;
		double weight = calcEventWeight( eventid);

		u32 itr = prglist;
		const ProgramTrigger* ptp;
		while (0!=(ptp=m_programTriggerList.nextptr( itr)))
		{
			// copy: pushes below may grow the pool the pointer refers into
			ProgramTrigger pt = *ptp;
			const Program& prg = m_programs.at( pt.programidx-1);
			u32 alt_eventid = getAltEventId( eventid, prg.triggerListIdx);
			if (!alt_eventid)
			{
				m_programTriggerList.push( new_prglist, pt);
			}
			else
			{
				double alt_weight = calcEventWeight( alt_eventid) * opt.weightFactor;
				if (!pt.past_eventid
				&&  opt.maxRange >= prg.positionRange
				&&  weight > alt_weight)
				{
					defineEventProgramAlt( alt_eventid, pt.programidx, eventid);
					getDelimTokenStopWordSet( prg.triggerListIdx);
				}
				else
				{
					m_programTriggerList.push( new_prglist, pt);
				}
			}
		}
		// NOTE: defineEventProgramAlt may have inserted into the unordered_map and rehashed it.
		// The reference keeps using `ei` afterwards (cpp:576-584); with libstdc++ an iterator is a
		// node pointer and survives a rehash, so the same sequence is executed here.
		m_programTriggerList.removeList( prglist);
		if (new_prglist != 0)
		{
			ei->second = new_prglist;
		}
		else
		{
			m_eventProgramTriggerMap.erase( ei);
		}
	}
}

// ------------------------------------------------------------------ StateMachine
// ruleMatcherAutomaton.cpp:589-602
StateMachine::StateMachine( const ProgramTable* programTable)
	:m_programTable(programTable),m_curpos(0)
	,m_nofProgramsInstalled(0),m_nofAltKeyProgramsInstalled(0),m_nofSignalsFired(0)
	,m_nofOpenPatterns(0.0),m_timestmp(0)
{
	std::memset( m_disposeWindow, 0, sizeof(m_disposeWindow));
}

// ruleMatcherAutomaton.cpp:672-677
u32 StateMachine::createRule( u32 expiryOrdpos)
{
	Rule r; r.actionSlotIdx = 0; r.eventTriggerListIdx = 0; r.eventDataReferenceIdx = 0; r.done = false; r.lastpos = expiryOrdpos;
	u32 rt = m_ruleTable.add( r);
	defineDisposeRule( expiryOrdpos, rt);
	return rt;
}

// ruleMatcherAutomaton.cpp:679-702
void StateMachine::deactivateRule( u32 ruleidx)
{
	Rule& rule = m_ruleTable.at( ruleidx);
	if (rule.isActive())
	{
		m_actionSlotTable.remove( rule.actionSlotIdx-1);
		rule.actionSlotIdx = 0;

		u32 itr = rule.eventTriggerListIdx;
		u32 trigger;
		while (m_eventTriggerList.next( itr, trigger))
		{
			if (trigger) m_eventTriggerTable.remove( trigger-1);
		}
		m_eventTriggerList.removeList( rule.eventTriggerListIdx);
		rule.eventTriggerListIdx = 0;

		if (rule.eventDataReferenceIdx)
		{
			disposeEventDataReference( rule.eventDataReferenceIdx);
			rule.eventDataReferenceIdx = 0;
		}
	}
}

// ruleMatcherAutomaton.cpp:704-708
void StateMachine::disposeRule( u32 ruleidx)
{
	deactivateRule( ruleidx);
	m_ruleTable.remove( ruleidx);
}

// ruleMatcherAutomaton.cpp:710-732.  Data references are handed around as (index+1), 0 = none.
void StateMachine::disposeEventDataReference( u32 ref)
{
	EventDataReference& r = m_eventDataReferenceTable.at( ref-1);
	if (r.referenceCount > 1)
	{
		--r.referenceCount;
	}
	else if (r.referenceCount == 1)
	{
		--r.referenceCount;
		if (r.eventItemListIdx) m_eventItemList.removeList( r.eventItemListIdx);
		m_eventDataReferenceTable.remove( ref-1);
	}
	else
	{
		throw std::runtime_error("illegal free of event data reference");
	}
}

// ruleMatcherAutomaton.cpp:734-738
void StateMachine::referenceEventData( u32 ref)
{
	++m_eventDataReferenceTable.at( ref-1).referenceCount;
}

// ruleMatcherAutomaton.cpp:740-748
void StateMachine::appendEventData( u32 ref, const EventItem& item)
{
	if (item.data.subdataref) referenceEventData( item.data.subdataref);
	m_eventItemList.push( m_eventDataReferenceTable.at( ref-1).eventItemListIdx, item);
}

// ruleMatcherAutomaton.cpp:750-753
u32 StateMachine::createEventData()
{
	EventDataReference r; r.eventItemListIdx = 0; r.referenceCount = 1;
	return m_eventDataReferenceTable.add( r) + 1;
}

// ruleMatcherAutomaton.cpp:755-770
void StateMachine::joinEventData( u32 dest, u32 src)
{
	u32 itr = m_eventDataReferenceTable.at( src-1).eventItemListIdx;
	const EventItem* ip;
	while (0!=(ip=m_eventItemList.nextptr( itr)))
	{
		EventItem item = *ip;	// copy: the push below may reallocate the pool
		if (item.data.subdataref) referenceEventData( item.data.subdataref);
		m_eventItemList.push( m_eventDataReferenceTable.at( dest-1).eventItemListIdx, item);
	}
}

// ruleMatcherAutomaton.cpp:772-979 (debug-trace branches omitted)
void StateMachine::fireSignal( u32 slotidx, const Trigger& trigger, const EventData& data,
				std::vector<u32>& disposeRuleList, std::vector<EventStruct>& followList)
{
	ActionSlot& slot = m_actionSlotTable.at( slotidx);
	Rule& rule = m_ruleTable.at( slot.rule);
	bool match = false;
	bool takeEventData = false;
	bool finished = false;
	++m_nofSignalsFired;

	switch ((SigType)trigger.sigtype)
	{
		case SigAny:
			takeEventData = true;
			if (slot.count > 0)
			{
				match = true;
				--slot.count;
				finished = (slot.count == 0);
				if (slot.end_ordpos < data.end_ordpos) slot.end_ordpos = data.end_ordpos;
			}
			break;
		case SigAnd:
			if (slot.count > 0)
			{
				if (!slot.value)
				{
					slot.value = data.start_ordpos;
					if (slot.end_ordpos > data.end_ordpos) slot.end_ordpos = data.end_ordpos;
				}
				if (slot.value == data.start_ordpos)
				{
					match = true;
					--slot.count;
					finished = (slot.count == 0);
					takeEventData = true;
				}
			}
			break;
		case SigSequence:
			if (trigger.sigval == slot.value && slot.end_ordpos <= data.start_ordpos)
			{
				slot.end_ordpos = data.end_ordpos;
				slot.value = trigger.sigval-1;
				if (slot.count > 0)
				{
					--slot.count;
					match = (slot.count == 0);
				}
				else
				{
					match = true;
				}
				finished = (slot.value == 0);
				takeEventData = true;
			}
			break;
		case SigSequenceImm:
			if (trigger.sigval == slot.value && slot.end_ordpos == data.start_ordpos)
			{
				slot.end_ordpos = data.end_ordpos;
				slot.value = trigger.sigval-1;
				if (slot.count > 0)
				{
					--slot.count;
					match = (slot.count == 0);
				}
				else
				{
					match = true;
				}
				finished = (slot.value == 0);
				takeEventData = true;
			}
			break;
		case SigWithin:
			if ((trigger.sigval & slot.value) != 0 && slot.end_ordpos <= data.start_ordpos)
			{
				slot.end_ordpos = data.end_ordpos;
				slot.value &= ~trigger.sigval;
				if (slot.count > 0)
				{
					--slot.count;
					match = (slot.count == 0);
				}
				else
				{
					match = true;
				}
				finished = (slot.value == 0);
				takeEventData = true;
			}
			break;
		case SigDel:
			slot.count = 0;
			slot.value = 0;
			disposeRuleList.push_back( slot.rule);
			return;
	}
	if (takeEventData)
	{
		if (trigger.variable)
		{
			EventItem item; item.variable = trigger.variable; item.data = data;
			if (!rule.eventDataReferenceIdx) rule.eventDataReferenceIdx = createEventData();
			appendEventData( rule.eventDataReferenceIdx, item);
		}
		else if (data.subdataref)
		{
			if (!rule.eventDataReferenceIdx) rule.eventDataReferenceIdx = createEventData();
			joinEventData( rule.eventDataReferenceIdx, data.subdataref);
		}
		if (slot.start_ordpos == 0)
		{
			slot.start_ordpos = data.start_ordpos;
			slot.start_origseg = data.start_origseg;
			slot.start_origpos = data.start_origpos;
		}
		else if (slot.start_ordpos > data.start_ordpos)
		{
			slot.start_ordpos = data.start_ordpos;
			if (slot.start_origseg > data.start_origseg || (slot.start_origseg == data.start_origseg && slot.start_origpos > data.start_origpos))
			{
				slot.start_origseg = data.start_origseg;
				slot.start_origpos = data.start_origpos;
			}
		}
	}
	if (match)
	{
		if (!rule.done)
		{
			if (slot.event)
			{
				EventStruct follow;
				follow.data = EventData( slot.start_origseg, slot.start_origpos, data.end_origseg, data.end_origpos, slot.start_ordpos, slot.end_ordpos, rule.eventDataReferenceIdx, slot.formatHandle);
				follow.eventid = slot.event;
				if (rule.eventDataReferenceIdx) referenceEventData( rule.eventDataReferenceIdx);
				followList.push_back( follow);
			}
			if (slot.resultHandle)
			{
				Result r;
				r.resultHandle = slot.resultHandle; r.formatHandle = slot.formatHandle;
				r.eventDataReferenceIdx = rule.eventDataReferenceIdx;
				r.start_ordpos = slot.start_ordpos; r.end_ordpos = slot.end_ordpos;
				r.start_origseg = slot.start_origseg; r.start_origpos = slot.start_origpos;
				r.end_origseg = data.end_origseg; r.end_origpos = data.end_origpos;
				m_results.push_back( r);
				if (rule.eventDataReferenceIdx) referenceEventData( rule.eventDataReferenceIdx);
			}
			rule.done = true;
		}
		if (finished) disposeRuleList.push_back( slot.rule);
	}
}

// ruleMatcherAutomaton.cpp:981-1064
void StateMachine::doTransition( u32 event, const EventData& data)
{
	m_nofOpenPatterns += m_eventTriggerTable.nofTriggers();

	std::vector<EventStruct> followList;
	{
		EventStruct first; first.data = data; first.eventid = event;
		followList.push_back( first);
	}
	if (followList[0].data.subdataref) referenceEventData( followList[0].data.subdataref);

	for (std::size_t ei=0; ei < followList.size(); ++ei)
	{
		std::vector<u32> triggers;
		std::vector<u32> disposeRuleList;
		EventStruct follow = followList[ ei];

		m_eventTriggerTable.getTriggers( triggers, follow.eventid);
		for (std::size_t ti=0; ti<triggers.size(); ++ti)
		{
			// the reference holds Trigger pointers collected before firing; nothing adds or
			// removes triggers inside this loop, so reading by index is the same thing
			Trigger trigger = m_eventTriggerTable.getTrigger( triggers[ ti]);
			fireSignal( trigger.slot, trigger, follow.data, disposeRuleList, followList);
		}
		installEventPrograms( follow.eventid, follow.data, followList, disposeRuleList);

		for (std::size_t di=0; di<disposeRuleList.size(); ++di) deactivateRule( disposeRuleList[ di]);

		if (m_programTable->isStopWord( follow.eventid))
		{
			EventLog lg; lg.data = follow.data; lg.timestmp = ++m_timestmp;
			m_stopWordsEventLogMap[ follow.eventid] = lg;
		}
		else if (follow.data.subdataref)
		{
			disposeEventDataReference( follow.data.subdataref);
		}
	}
}

// ruleMatcherAutomaton.cpp:1066-1082
void StateMachine::defineDisposeRule( u32 pos, u32 ruleidx)
{
	if (pos < m_curpos) throw std::runtime_error("illegal definition of dispose rule (position smaller than current)");
	if (pos < m_curpos + DisposeWindowSize)
	{
		m_disposeRuleList.push( m_disposeWindow[ pos % DisposeWindowSize], ruleidx);
	}
	else
	{
		DisposeEvent de; de.pos = pos; de.idx = ruleidx;
		m_ruleDisposeQueue.push_back( de);
		std::push_heap( m_ruleDisposeQueue.begin(), m_ruleDisposeQueue.end());
	}
}

// ruleMatcherAutomaton.cpp:1084-1135
void StateMachine::setCurrentPos( u32 pos)
{
	if (pos < m_curpos) throw std::runtime_error("illegal definition of current pos (positions not ascending)");
	if (m_curpos == pos) return;

	std::size_t wcnt=0;
	for (; wcnt < DisposeWindowSize && m_curpos < pos; ++wcnt,++m_curpos)
	{
		u32 widx = m_curpos % DisposeWindowSize;
		if (widx == 0)
		{
			while (!m_ruleDisposeQueue.empty() && m_ruleDisposeQueue.front().pos < m_curpos + DisposeWindowSize)
			{
				wcnt = 0;
				m_disposeRuleList.push( m_disposeWindow[ m_ruleDisposeQueue.front().pos % DisposeWindowSize], m_ruleDisposeQueue.front().idx);
				std::pop_heap( m_ruleDisposeQueue.begin(), m_ruleDisposeQueue.end());
				m_ruleDisposeQueue.pop_back();
			}
		}
		u32 rulelist = m_disposeWindow[ widx];
		if (rulelist)
		{
			// the reference walks the list while disposing; the list nodes are only freed
			// afterwards... they are in fact never freed there (window slot is just zeroed), which
			// only leaks pool nodes and has no observable effect.
			u32 itr = rulelist;
			u32 ruleidx;
			while (m_disposeRuleList.next( itr, ruleidx)) disposeRule( ruleidx);
			m_disposeWindow[ widx] = 0;
		}
	}
	if (m_curpos < pos)
	{
		m_curpos = pos;
		while (!m_ruleDisposeQueue.empty() && m_ruleDisposeQueue.front().pos < m_curpos)
		{
			disposeRule( m_ruleDisposeQueue.front().idx);
			std::pop_heap( m_ruleDisposeQueue.begin(), m_ruleDisposeQueue.end());
			m_ruleDisposeQueue.pop_back();
		}
	}
}

// ruleMatcherAutomaton.cpp:1137-1157
void StateMachine::installEventPrograms( u32 event, const EventData& data,
				std::vector<EventStruct>& followList, std::vector<u32>& disposeRuleList)
{
	u32 lst = m_programTable->getEventProgramList( event);
	const ProgramTrigger* pt;
	while (0!=(pt=m_programTable->nextProgramPtr( lst)))
	{
		installProgram( event, *pt, data, followList, disposeRuleList);
	}
}

// ruleMatcherAutomaton.cpp:1159-1166
static bool triggerDefNeedsInstall( const TriggerDef& td, const ActionSlot& slot)
{
	return ((SigType)td.sigtype == SigAny && slot.count > 1);
}

// ruleMatcherAutomaton.cpp:1168-1270
void StateMachine::installProgram( u32 keyevent, const ProgramTrigger& programTrigger, const EventData& data,
				std::vector<EventStruct>& followList, std::vector<u32>& disposeRuleList)
{
	const Program& program = m_programTable->program( programTrigger.programidx);
	if (data.start_ordpos + program.positionRange < m_curpos) return;

	u32 ruleidx = createRule( data.start_ordpos + program.positionRange);
	{
		ActionSlot s;
		s.value = program.slotDef.initsigval; s.count = (uint16_t)program.slotDef.initcount;
		s.event = program.slotDef.event; s.rule = ruleidx;
		s.resultHandle = program.slotDef.resultHandle; s.formatHandle = program.slotDef.formatHandle;
		s.start_ordpos = 0; s.end_ordpos = 0; s.start_origseg = 0; s.start_origpos = 0;
		m_ruleTable.at( ruleidx).actionSlotIdx = 1 + m_actionSlotTable.add( s);
	}
	u32 slotidx = m_ruleTable.at( ruleidx).actionSlotIdx - 1;

	enum {MaxNofKeyTriggerDefs=32};
	TriggerDef keyTriggerDef[ MaxNofKeyTriggerDefs];
	std::size_t nofKeyTriggerDef = 0;
	bool hasKeyEvent = false;
	u32 itr = program.triggerListIdx;
	const TriggerDef* td;
	while (0!=(td=m_programTable->triggerList().nextptr( itr)))
	{
		bool doInstall = false;
		if (keyevent == td->event)
		{
			if (nofKeyTriggerDef < MaxNofKeyTriggerDefs)
			{
				keyTriggerDef[ nofKeyTriggerDef++] = *td;
			}
			else
			{
				throw std::runtime_error("pattern with too many identical key events defined");
			}
			if (td->isKeyEvent && !hasKeyEvent)
			{
				hasKeyEvent = true;
				if (triggerDefNeedsInstall( *td, m_actionSlotTable.at( slotidx))) doInstall = true;
			}
			else if ((SigType)td->sigtype == SigDel)
			{
				if (triggerDefNeedsInstall( *td, m_actionSlotTable.at( slotidx))) doInstall = true;
			}
			else
			{
				doInstall = true;
			}
		}
		else
		{
			doInstall = true;
		}
		if (doInstall)
		{
			Trigger t; t.slot = slotidx; t.sigtype = td->sigtype; t.sigval = td->sigval; t.variable = td->variable;
			u32 eventTrigger = m_eventTriggerTable.add( td->event, t);
			m_eventTriggerList.push( m_ruleTable.at( ruleidx).eventTriggerListIdx, eventTrigger+1);
		}
	}
	m_nofProgramsInstalled += 1;

	if (programTrigger.past_eventid)
	{
		m_nofAltKeyProgramsInstalled += 1;
		replayPastEvent( programTrigger.past_eventid, ruleidx, program.positionRange);
	}
	if (nofKeyTriggerDef && m_ruleTable.at( ruleidx).actionSlotIdx)
	{
		for (std::size_t ki=0; ki < nofKeyTriggerDef; ++ki)
		{
			Trigger keyTrigger; keyTrigger.slot = slotidx; keyTrigger.sigtype = keyTriggerDef[ki].sigtype;
			keyTrigger.sigval = keyTriggerDef[ki].sigval; keyTrigger.variable = keyTriggerDef[ki].variable;
			fireSignal( slotidx, keyTrigger, data, disposeRuleList, followList);
		}
	}
}

// ruleMatcherAutomaton.cpp:1272-1334
void StateMachine::replayPastEvent( u32 eventid, u32 ruleidx, u32 positionRange)
{
	std::map<u32,EventLog>::const_iterator ei = m_stopWordsEventLogMap.find( eventid);
	if (ei != m_stopWordsEventLogMap.end() && ei->second.data.start_ordpos + positionRange >= m_curpos)
	{
		u32 slotidx = m_ruleTable.at( ruleidx).actionSlotIdx-1;
		std::vector<EventStruct> followList;
		std::vector<u32> disposeRuleList;
		std::vector<u32> delEventList;

		u32 itr = m_ruleTable.at( ruleidx).eventTriggerListIdx;
		u32 trigger;
		while (m_eventTriggerList.next( itr, trigger))
		{
			Trigger tp = m_eventTriggerTable.getTrigger( trigger-1);
			u32 trigger_eventid = m_eventTriggerTable.getTriggerEventId( trigger-1);
			if ((SigType)tp.sigtype == SigDel) delEventList.push_back( trigger_eventid);
			if (eventid == trigger_eventid)
			{
				fireSignal( slotidx, tp, ei->second.data, disposeRuleList, followList);
			}
		}
		for (std::size_t li=0; li<delEventList.size(); ++li)
		{
			std::map<u32,EventLog>::const_iterator stopwi = m_stopWordsEventLogMap.find( delEventList[ li]);
			if (stopwi != m_stopWordsEventLogMap.end() && stopwi->second.timestmp > ei->second.timestmp)
			{
				deactivateRule( ruleidx);
				break;
			}
		}
		for (std::size_t di=0; di<disposeRuleList.size(); ++di) deactivateRule( disposeRuleList[ di]);
		if (!followList.empty())
		{
			throw std::runtime_error("internal: encountered past trigger with follow");
		}
	}
}

// ------------------------------------------------------------------ MatcherInstance
// patternMatcher.cpp:361-364
void MatcherInstance::defineTermFrequency( u32 termid, double df)
{
	m_programTable.defineEventFrequency( eventHandle( TermEvent, termid), df);
}

// patternMatcher.cpp:366-375
void MatcherInstance::pushTerm( u32 termid)
{
	m_stack.push_back( StackElement( eventHandle( TermEvent, termid)));
}

// patternMatcher.cpp:377-508
void MatcherInstance::pushExpression( JoinOp joinop, std::size_t argc, u32 range, u32 cardinality)
{
	if (argc > m_stack.size()) throw std::runtime_error("expression references more arguments than nodes on the stack");
	u32 slot_initsigval = 0;
	u32 slot_initcount = cardinality ? cardinality : (u32)argc;
	u32 slot_event = eventHandle( ExpressionEvent, ++m_expression_event_cnt);
	SigType slot_sigtype = SigAny;

	switch (joinop)
	{
		case OpSequence:	slot_sigtype = SigSequence; slot_initsigval = (u32)argc; break;
		case OpSequenceImm:	slot_sigtype = SigSequenceImm; slot_initsigval = (u32)argc; break;
		case OpSequenceStruct:	slot_sigtype = SigSequence; slot_initsigval = (u32)argc-1; --slot_initcount; break;
		case OpWithin:
			slot_sigtype = SigWithin;
			if (argc > 32) throw std::runtime_error("operator 'within': number of arguments out of range");
			slot_initsigval = 0xffFFffFF;
			break;
		case OpWithinStruct:
			slot_sigtype = SigWithin;
			if (argc > 32) throw std::runtime_error("operator 'within_struct': number of arguments out of range");
			slot_initsigval = 0xffFFffFF;
			--slot_initcount;
			break;
		case OpAny:		slot_sigtype = SigAny; slot_initcount = cardinality ? cardinality : 1; break;
		case OpAnd:		slot_sigtype = SigAnd; break;
	}
	ActionSlotDef def; def.initsigval = slot_initsigval; def.initcount = slot_initcount; def.event = slot_event;
	def.resultHandle = 0; def.formatHandle = 0;
	u32 program = m_programTable.createProgram( range, def);

	for (std::size_t ai=0; ai != argc; ++ai)
	{
		bool isKeyEvent = false;
		u32 trigger_sigval = 0;
		SigType trigger_sigtype = slot_sigtype;
		switch (joinop)
		{
			case OpSequenceStruct:
				if (ai == 0) trigger_sigtype = SigDel;
				else { trigger_sigval = (u32)(argc-ai); isKeyEvent = (ai == 1); }
				break;
			case OpWithinStruct:
				if (ai == 0) trigger_sigtype = SigDel;
				else { trigger_sigval = 1u << (argc-ai); isKeyEvent = true; }
				break;
			case OpSequence:
				trigger_sigval = (u32)(argc-ai); isKeyEvent = (ai == 0);
				break;
			case OpSequenceImm:
				if (ai == 0) trigger_sigtype = SigSequence;
				trigger_sigval = (u32)(argc-ai); isKeyEvent = (ai == 0);
				break;
			case OpWithin:
				trigger_sigval = 1u << (argc-ai-1); isKeyEvent = true;
				break;
			case OpAny:
			case OpAnd:
				isKeyEvent = true;
				break;
		}
		const StackElement& elem = m_stack[ m_stack.size() - argc + ai];
		m_programTable.createTrigger( program, elem.eventid, isKeyEvent, trigger_sigtype, trigger_sigval, elem.variable);
	}
	m_programTable.doneProgram( program);
	m_stack.erase( m_stack.end() - argc, m_stack.end());
	m_stack.push_back( StackElement( slot_event, program));
}

// patternMatcher.cpp:510-520
void MatcherInstance::pushPattern( const std::string& name)
{
	m_stack.push_back( StackElement( eventHandle( ReferenceEvent, m_patternMap.getOrCreate( name))));
}

// patternMatcher.cpp:522-543
void MatcherInstance::attachVariable( const std::string& name)
{
	if (m_stack.empty()) throw std::runtime_error("illegal operation attach variable when no node on the stack");
	StackElement& elem = m_stack.back();
	if (elem.variable) throw std::runtime_error("more than one variable assignment to a node");
	elem.variable = m_variableMap.getOrCreate( name);
}

// patternMatcher.cpp:545-584
void MatcherInstance::definePattern( const std::string& name, const std::string& formatstring, bool visible)
{
	if (m_stack.empty()) throw std::runtime_error("illegal operation close pattern when no node on the stack");
	StackElement& elem = m_stack.back();
	u32 resultHandle = m_patternMap.getOrCreate( name);
	u32 resultEvent = eventHandle( ReferenceEvent, resultHandle);
	u32 program = elem.program;
	u32 formatHandle = 0;
	if (!formatstring.empty()) formatHandle = ++m_nofFormats;
	if (!program)
	{
		ActionSlotDef def; def.initsigval = 0; def.initcount = 1; def.event = resultEvent;
		def.resultHandle = resultHandle; def.formatHandle = formatHandle;
		program = m_programTable.createProgram( 0, def);
		m_programTable.createTrigger( program, elem.eventid, true, SigAny, 0, elem.variable);
		m_programTable.doneProgram( program);
	}
	else if (elem.variable)
	{
		throw std::runtime_error("variable assignments only allowed to subexpressions of pattern");
	}
	m_programTable.defineProgramResult( program, resultEvent, visible ? resultHandle : 0, formatHandle);
}

static bool ieq( const std::string& a, const char* b)
{
	std::size_t n = std::strlen( b);
	if (a.size() != n) return false;
	for (std::size_t i=0; i<n; ++i) if ((a[i]|32) != (b[i]|32)) return false;
	return true;
}

// patternMatcher.cpp:614-644
void MatcherInstance::defineOption( const std::string& name, double value)
{
	if (ieq( name, "stopwordOccurrenceFactor")) m_popt.stopwordOccurrenceFactor = (float)value;
	else if (ieq( name, "weightFactor")) m_popt.weightFactor = (float)value;
	else if (ieq( name, "maxRange")) m_popt.maxRange = (unsigned int)(value + std::numeric_limits<double>::epsilon());
	else if (ieq( name, "maxResultSize")) m_maxResultSize = (unsigned int)(value + std::numeric_limits<double>::epsilon());
	else if (ieq( name, "exclusive")) m_exclusive = true;
	else throw std::runtime_error("unknown token pattern match option: '" + name + "'");
}

// patternMatcher.cpp:646-671
bool MatcherInstance::compile()
{
	m_programTable.optimize( m_popt);
	return true;
}

// ------------------------------------------------------------------ MatcherContext
MatcherContext::MatcherContext( const MatcherInstance* inst)
	:m_inst(inst),m_sm(new StateMachine( &inst->programTable())),m_nofEvents(0),m_curPosition(0){}
MatcherContext::~MatcherContext() {delete m_sm;}

// patternMatcher.cpp:320-331
void MatcherContext::reset()
{
	StateMachine* n = new StateMachine( &m_inst->programTable());
	delete m_sm; m_sm = n; m_nofEvents = 0; m_curPosition = 0;
}

// patternMatcher.cpp:131-162 (the three range checks sit in the else-if chain, B.8)
void MatcherContext::putInput( const Lexem& term)
{
	const u32 lim = (u32)std::numeric_limits<int32_t>::max();
	if (m_curPosition > term.ordpos)
	{
		throw std::runtime_error("term events not fed in ascending order");
	}
	else if (m_curPosition < term.ordpos)
	{
		m_sm->setCurrentPos( m_curPosition = term.ordpos);
	}
	else if (term.origsize >= lim) throw std::runtime_error("term event orig size out of range");
	else if (term.origseg >= lim) throw std::runtime_error("term event orig segment number out of range");
	else if (term.origpos >= lim) throw std::runtime_error("term event orig segment byte position out of range");

	u32 eventid = eventHandle( TermEvent, term.id);
	EventData data( term.origseg, term.origpos, term.origseg, term.origpos + term.origsize, term.ordpos, term.ordpos+1, 0, 0);
	m_sm->doTransition( eventid, data);
	++m_nofEvents;
}

// patternMatcher.cpp:164-190
void MatcherContext::gatherResultItems( std::vector<ResultItem>& out, u32 dataref) const
{
	u32 itr = m_sm->getEventDataItemListIdx( dataref-1);
	const EventItem* item;
	while (0!=(item=m_sm->nextResultItem( itr)))
	{
		ResultItem ri;
		ri.variable = item->variable;
		ri.start_ordpos = item->data.start_ordpos; ri.end_ordpos = item->data.end_ordpos;
		ri.start_origseg = item->data.start_origseg; ri.start_origpos = item->data.start_origpos;
		ri.end_origseg = item->data.end_origseg; ri.end_origpos = item->data.end_origpos;
		ri.formatHandle = item->data.formathandle;
		if (item->data.formathandle && item->data.subdataref)
		{
			gatherResultItems( ri.args, item->data.subdataref);	// :175-179: arguments of the item's format string
		}
		out.push_back( ri);
		if (item->data.subdataref && !item->data.formathandle)
		{
			gatherResultItems( out, item->data.subdataref);
		}
	}
}

// patternMatcher.cpp:192-246
std::vector<bool> MatcherContext::getCoveredFlags( const std::vector<Result>& results) const
{
	std::vector<bool> rt( results.size(), false);
	u32 maxResultSize = m_inst->maxResultSize();
	for (std::size_t ai=0; ai != results.size(); ++ai)
	{
		const Result& result = results[ ai];
		for (std::size_t ni=ai; ni != results.size(); ++ni)
		{
			const Result& fr = results[ ni];
			if (fr.start_origseg > result.end_origseg
			||  fr.start_origpos >= result.end_origpos + maxResultSize)
			{
				break;
			}
			bool differ = (fr.end_origseg != result.end_origseg || fr.end_origpos != result.end_origpos
					|| fr.start_origseg != result.start_origseg || fr.start_origpos != result.start_origpos);
			if (fr.start_origseg <= result.start_origseg && fr.start_origpos <= result.start_origpos
			&&  fr.end_origseg >= result.end_origseg && fr.end_origpos >= result.end_origpos)
			{
				if (differ) rt[ ai] = true;
			}
			if (fr.start_origseg >= result.start_origseg && fr.start_origpos >= result.start_origpos
			&&  fr.end_origseg <= result.end_origseg && fr.end_origpos <= result.end_origpos)
			{
				if (differ) rt[ ni] = true;
			}
		}
	}
	return rt;
}

// patternMatcher.cpp:248-301
std::vector<MatchResult> MatcherContext::fetchResults() const
{
	const std::vector<Result>& results = m_sm->results();
	std::vector<bool> eliminate;
	if (m_inst->exclusive()) eliminate = getCoveredFlags( results);
	std::vector<MatchResult> rt;
	for (std::size_t ai=0; ai != results.size(); ++ai)
	{
		if (m_inst->exclusive() && eliminate[ ai]) continue;
		const Result& r = results[ ai];
		MatchResult m;
		m.resultHandle = r.resultHandle;
		m.start_ordpos = r.start_ordpos; m.end_ordpos = r.end_ordpos;
		m.start_origseg = r.start_origseg; m.start_origpos = r.start_origpos;
		m.end_origseg = r.end_origseg; m.end_origpos = r.end_origpos;
		m.formatHandle = r.formatHandle;
		// :253-266: with a format handle the gathered list feeds the format string, without one it is the item list
		if (r.eventDataReferenceIdx) gatherResultItems( m.items, r.eventDataReferenceIdx);
		rt.push_back( m);
	}
	return rt;
}
