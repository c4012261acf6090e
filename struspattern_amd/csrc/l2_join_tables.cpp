// Join prototype (l2_join.h): which rule sets it takes, and its tables.
#include "l2_join.h"
#include "l2_compile.hpp"
#include <map>
#include <set>
#include <string>
#include <vector>

namespace spa {

// returns the reason why the rule set cannot run in join mode (empty = it can)
std::string buildJoinTables( const FlatTables& ft, std::vector<JoinKey>& keytab, std::vector<JoinRule>& rules, uint32_t& maxRange)
{
	std::map<std::pair<uint32_t,uint32_t>,std::vector<JoinRule> > byPair;
	maxRange = 0;
	for (size_t ki=0; ki<ft.keylist.size(); ++ki) if (ft.keylist[ ki].pastEvent) return "a program with an alternative key (compile without optimize)";
	std::set<uint32_t> listened;
	for (size_t i=0; i<ft.trigdefs.size(); ++i) listened.insert( ft.trigdefs[ i].event);
	for (size_t pi=0; pi<ft.programs.size(); ++pi)
	{
		const DevProgram& p = ft.programs[ pi];
		if (p.event && listened.count( p.event)) return "a program whose result another program listens to";
		if (!p.resultHandle) return "a program without result";
		if (p.trigCount != 2 || p.initsigval != 2 || p.initcount != 2) return "a program that is not a two-term sequence";
		uint32_t first = 0, second = 0;
		for (uint32_t t=0; t<2; ++t)
		{
			const DevTrigDef& td = ft.trigdefs[ p.trigBegin + t];
			if ((td.flags & 0xF) != SIG_SEQUENCE) return "a program that is not a plain two-term sequence";	// (variables are accepted, captured items are not produced)
			if (td.sigval == 2 && (td.flags & 0x100)) first = td.event;
			else if (td.sigval == 1 && !(td.flags & 0x100)) second = td.event;
		}
		if (!first || !second) return "a program that is not a plain two-term sequence";
		JoinRule r; r.range = p.positionRange; r.resultHandle = p.resultHandle; r.formatHandle = p.formatHandle; r._pad = 0;
		if (r.range > maxRange) maxRange = r.range;
		byPair[ std::make_pair( first, second)].push_back( r);
	}
	if (byPair.empty()) return "no program";
	size_t size = 1;
	while (size < byPair.size()*2+1) size <<= 1;
	JoinKey none; none.first = 0; none.second = 0; none.begin = 0; none.count = 0;
	keytab.assign( size, none); rules.clear();
	for (std::map<std::pair<uint32_t,uint32_t>,std::vector<JoinRule> >::const_iterator it=byPair.begin(); it!=byPair.end(); ++it)
	{
		JoinKey k; k.first = it->first.first; k.second = it->first.second; k.begin = (uint32_t)rules.size(); k.count = (uint32_t)it->second.size();
		rules.insert( rules.end(), it->second.begin(), it->second.end());
		size_t slot = joinHash( k.first, k.second) & (size-1);
		while (keytab[ slot].first) slot = (slot+1) & (size-1);
		keytab[ slot] = k;
	}
	return std::string();
}

} // namespace
