// Join prototype (l2_join.h): which rule sets it takes, and its tables.
#include "l2_join.h"
#include "l2_compile.hpp"
#include <map>
#include <set>
#include <string>
#include <vector>

namespace spa {

// returns the reason why the rule set cannot run in join mode (empty = it can)
std::string buildJoinTables( const FlatTables& ft, std::vector<JoinKey>& keytab, std::vector<JoinRule>& rules, std::vector<uint32_t>& filter, uint32_t& maxRange, uint32_t& delimiter)
{
	std::map<std::pair<uint32_t,uint32_t>,std::vector<JoinRule> > byPair;
	maxRange = 0; delimiter = 0;
	std::set<uint32_t> listened;
	for (size_t i=0; i<ft.trigdefs.size(); ++i) listened.insert( ft.trigdefs[ i].event);
	for (size_t pi=0; pi<ft.programs.size(); ++pi)
	{
		const DevProgram& p = ft.programs[ pi];
		if (p.event && listened.count( p.event)) return "a program whose result another program listens to";
		if (!p.resultHandle) return "a program without result";
	}
	for (size_t pi=0; pi<ft.programs.size(); ++pi)
	{
		const DevProgram& p = ft.programs[ pi];
		// the two terms and the delimiter of the program.  Which of the terms the optimizer made the key does not matter here:
		// the entries follow the program's meaning -- sequence: first term -> second; within: either way round; any: each term
		// alone -- which is what the key lists of the unoptimized automaton give (a program with two equal terms counts twice
		// there, and twice here).
		const DevTrigDef* term[ 2] = {0, 0}; const DevTrigDef* del = 0;
		unsigned nterm = 0;
		for (uint32_t t=0; t<p.trigCount; ++t)
		{
			const DevTrigDef& td = ft.trigdefs[ p.trigBegin + t];
			if ((td.flags & 0xF) == SIG_DEL) { if (del) return "a program with two delimiters"; del = &td; }
			else { if (nterm == 2) return "a program with more than two terms"; term[ nterm++] = &td; }
		}
		if (nterm != 2) return "a program that has not two terms";
		if (term[ 0]->variable > 255 || term[ 1]->variable > 255) return "a variable id beyond 255";
		const uint32_t sigtype = term[ 0]->flags & 0xF;
		if ((term[ 1]->flags & 0xF) != sigtype) return "a program with mixed signals";
		JoinRule r; r.range = p.positionRange; r.resultHandle = p.resultHandle; r.formatHandle = p.formatHandle; r.flags = 0;
		if (del)
		{
			if (delimiter && delimiter != del->event) return "more than one delimiter event";
			delimiter = del->event; r.flags |= JOIN_STRUCT;
			if (term[ 0]->event == delimiter || term[ 1]->event == delimiter) return "the delimiter as a term";
		}
		if (r.range > maxRange) maxRange = r.range;
		if (sigtype == SIG_ANY)
		{
			if (p.initcount != 1 || del) return "an `any` program with a cardinality or a delimiter";
			// (two equal terms: both triggers of both instances take the lexem, term[0] in table order first -- listed last)
			const bool same = term[ 0]->event == term[ 1]->event;
			for (int t=0; t<2; ++t)
			{
				JoinRule rt = r; rt.flags |= same ? ((term[ 1]->variable << 16) | (term[ 0]->variable << 8)) : (term[ t]->variable << 16);
				byPair[ std::make_pair( (uint32_t)JOIN_SELF, term[ t]->event)].push_back( rt);
			}
		}
		else if (sigtype == SIG_SEQUENCE)
		{
			if (p.initcount != 2 || p.initsigval != 2) return "a sequence with a cardinality";
			const int first = term[ 0]->sigval == 2 ? 0 : 1;
			if (term[ first]->sigval != 2 || term[ 1-first]->sigval != 1) return "a sequence with unexpected signal values";
			r.flags |= (term[ first]->variable << 8) | (term[ 1-first]->variable << 16);
			byPair[ std::make_pair( term[ first]->event, term[ 1-first]->event)].push_back( r);
		}
		else if (sigtype == SIG_WITHIN)
		{
			if (p.initcount != 2) return "a within with a cardinality";
			// (two equal terms: the key lexem is taken by the trigger installed first, term[0] in table order, in both instances)
			const bool same = term[ 0]->event == term[ 1]->event;
			for (int t=0; t<2; ++t)
			{
				const int a = same ? 0 : t;
				JoinRule rt = r; rt.flags |= (term[ a]->variable << 8) | (term[ 1-a]->variable << 16);
				byPair[ std::make_pair( term[ t]->event, term[ 1-t]->event)].push_back( rt);
			}
		}
		else return "a program that is neither sequence, within nor any";
	}
	if (byPair.empty()) return "no program";
	size_t size = 1;
	while (size < byPair.size()*2+1) size <<= 1;
	JoinKey none; none.first = 0; none.second = 0; none.begin = 0; none.count = 0;
	keytab.assign( size, none); rules.clear();
	filter.assign( JOIN_FILTER_WORDS, 0);
	for (std::map<std::pair<uint32_t,uint32_t>,std::vector<JoinRule> >::const_iterator it=byPair.begin(); it!=byPair.end(); ++it)
	{
		JoinKey k; k.first = it->first.first; k.second = it->first.second; k.begin = (uint32_t)rules.size(); k.count = (uint32_t)it->second.size();
		rules.insert( rules.end(), it->second.begin(), it->second.end());
		size_t slot = joinHash( k.first, k.second) & (size-1);
		while (keytab[ slot].first) slot = (slot+1) & (size-1);
		keytab[ slot] = k;
		const uint32_t bit = (joinHash( k.first, k.second) >> 8) % (JOIN_FILTER_WORDS*32u);
		filter[ bit >> 5] |= 1u << (bit & 31u);
	}
	return std::string();
}

} // namespace
