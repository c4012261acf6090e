// Host side of the fast tier of level 2: decides whether a compiled rule set is FLAT (l2_fast.h) and
// builds the one-line-per-install table the fast kernel reads.
#include "l2_fast.h"
#include "l2_compile.hpp"
#include <map>
#include <set>
#include <cstring>

namespace spa {

static_assert( sizeof(FastKeyInst) == 64, "one install record per 64-byte line");
static_assert( sizeof(FastKeyEntry) == sizeof(DevKeyEntry), "the fast kernel reads the general key table in place");

// Why a rule set is not eligible (diagnostics, SPA_L2_VERBOSE), empty when it is.
std::string buildFastTables( const FlatTables& ft, std::vector<FastKeyInst>& out)
{
	out.clear();
	if (ft.nofStopWords > FAST_MAXSTOP) return "more than 64 stop words";
	if (ft.keylist.size() >= (1u<<24)) return "more than 2^24 key list entries";
	// events somebody waits for or is keyed by
	std::set<uint32_t> listened;
	std::map<uint32_t,uint32_t> stopIdxOf;
	for (size_t i=0; i<ft.trigdefs.size(); ++i) listened.insert( ft.trigdefs[ i].event);
	for (size_t i=0; i<ft.keytab.size(); ++i)
	{
		const DevKeyEntry& e = ft.keytab[ i];
		if (!e.event) continue;
		if (e.listCount) listened.insert( e.event);
		if (e.stopIdx) stopIdxOf[ e.event] = e.stopIdx;
	}
	for (size_t pi=0; pi<ft.programs.size(); ++pi)
	{
		const DevProgram& p = ft.programs[ pi];
		if (p.trigCount == 0) continue;				// (placeholder of an empty table)
		if (p.trigCount > 3) return "program with more than 3 triggers";
		if (p.positionRange > 63) return "position range above 63 (far-expiry queue)";
		if ((p.initcount & 0xFFFFu) > 31) return "cardinality above 31";
		if (p.event && listened.count( p.event)) return "nested expressions (a rule listens to another rule's result)";
		bool within = false, seq = false;
		for (uint32_t j=0; j<p.trigCount; ++j)
		{
			const DevTrigDef& t = ft.trigdefs[ p.trigBegin + j];
			const uint32_t sigtype = t.flags & 15u;
			if (sigtype == SIG_AND) return "operator 'and'";
			if (sigtype == SIG_WITHIN) within = true;
			if (sigtype == SIG_SEQUENCE || sigtype == SIG_SEQUENCE_IMM) seq = true;
			if (sigtype != SIG_DEL && sigtype != SIG_ANY && (t.sigval > 15u || t.sigval == 0)) return "signal value outside 1..15";
			if (t.variable > 255u) return "more than 255 variables";
		}
		if (within && seq) return "program mixing sequence and within triggers";
		if (within && p.initsigval != 0xFFFFFFFFu) return "within program with a partial initial mask";
		if (seq && p.initsigval > 15u) return "sequence of more than 15 elements";
	}
	out.resize( ft.keylist.size());
	std::memset( out.data(), 0, out.size()*sizeof(FastKeyInst));
	for (size_t ei=0; ei<ft.keytab.size(); ++ei)
	{
		const DevKeyEntry& e = ft.keytab[ ei];
		if (!e.event) continue;
		for (uint32_t k=0; k<e.listCount; ++k)
		{
			const DevKeyRef& ref = ft.keylist[ e.listBegin + k];
			const DevProgram& p = ft.programs[ ref.program];
			FastKeyInst& ki = out[ e.listBegin + k];
			ki.resultHandle = p.resultHandle; ki.formatHandle = p.formatHandle;
			ki.pastEvent = ref.pastEvent; ki.program = ref.program;
			bool within = false;
			for (uint32_t j=0; j<p.trigCount; ++j) if ((ft.trigdefs[ p.trigBegin + j].flags & 15u) == SIG_WITHIN) within = true;
			const uint32_t count = p.initcount & 0xFFFFu;		// ActionSlot::count is 16 bit (src/ruleMatcherAutomaton.hpp:98)
			const uint32_t value = within ? 0xFu : (p.initsigval & 0xFu);
			if (ref.pastStopIdx > FKI_PASTSTOP_MASK) return "stop word index out of range";
			ki.meta = value | (count << FKI_COUNT_SHIFT) | (p.positionRange << FKI_RANGE_SHIFT) | (p.trigCount << FKI_NTRIG_SHIFT)
				| (ref.pastStopIdx << FKI_PASTSTOP_SHIFT);
			// which templates are installed and which fire at once is decided by the key event alone
			// (triggerDefNeedsInstall, src/ruleMatcherAutomaton.cpp:1159-1166, :1207-1236)
			bool hasKey = false;
			for (uint32_t j=0; j<p.trigCount; ++j)
			{
				const DevTrigDef& t = ft.trigdefs[ p.trigBegin + j];
				const uint32_t sigtype = t.flags & 15u;
				bool doInstall, key = false;
				if (t.event == e.event)
				{
					key = true;
					const bool needs = (sigtype == SIG_ANY && count > 1);
					if ((t.flags & 0x100u) && !hasKey) { hasKey = true; doInstall = needs; }
					else if (sigtype == SIG_DEL) doInstall = needs;
					else doInstall = true;
				}
				else doInstall = true;
				uint32_t a = t.event;				// evhash of src/ruleMatcherAutomaton.cpp:34-40
				a += ~(a>>5); a += (a<<3); a ^= (a>>4);
				uint32_t delStop = 0;
				if (sigtype == SIG_DEL)
				{
					std::map<uint32_t,uint32_t>::const_iterator si = stopIdxOf.find( t.event);
					if (si != stopIdxOf.end()) delStop = si->second;
				}
				ki.trig[ j].event = t.event;
				ki.trig[ j].info = (t.sigval & FTI_SIGVAL_MASK) | (sigtype << FTI_SIGTYPE_SHIFT) | (doInstall ? (uint32_t)FTI_INSTALL : 0u) | (key ? (uint32_t)FTI_KEY : 0u)
					| ((a & 15u) << FTI_BUCKET_SHIFT) | (t.variable ? (uint32_t)FTI_HASVAR : 0u) | (delStop << FTI_DELSTOP_SHIFT) | (t.variable << FTI_VAR_SHIFT);
			}
		}
	}
	return std::string();
}

// Spill layout and the LDS regions of the 16 trigger buckets for the capacities of a kernel instance (R rules,
// T bucket entries in LDS; maxRules rule ids in all).  A bucket's share of T follows the number of trigger
// templates the rule set can install into it (the bucket of the structure delimiter holds a Del trigger of
// every *_struct rule instance); what does not fit spills per bucket.
void layoutFast( FastSpillLayout& S, uint32_t bucketMeta[16], uint32_t& expShift, const std::vector<FastKeyInst>& keyinst, uint32_t R, uint32_t T, uint32_t maxRules, uint32_t maxStaged)
{
	uint32_t maxRange = 0;
	for (size_t i=0; i<keyinst.size(); ++i) { const uint32_t rg = (keyinst[ i].meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK; if (rg > maxRange) maxRange = rg; }
	expShift = 0;
	while ((1u << expShift) < maxRange + 1u) ++expShift;		// one expiry row per position a live rule can expire at
	if (maxRules < R) maxRules = R;
	uint64_t weight[ 16], total = 0;
	for (int b=0; b<16; ++b) weight[ b] = 1;
	for (size_t i=0; i<keyinst.size(); ++i)
	{
		for (int j=0; j<3; ++j)
		{
			const uint32_t info = keyinst[ i].trig[ j].info;
			if (info & FTI_INSTALL) weight[ (info >> FTI_BUCKET_SHIFT) & 15u] += 1;
		}
	}
	for (int b=0; b<16; ++b) total += weight[ b];
	const uint32_t floorCap = 8;
	uint32_t base = 0;
	for (int b=0; b<16; ++b)
	{
		uint32_t cap = floorCap + (uint32_t)((uint64_t)(T - 16*floorCap) * weight[ b] / total);
		if (cap > 0xFFFu) cap = 0xFFFu;
		bucketMeta[ b] = base | (cap << 16);
		base += cap;
	}
	uint32_t w = 0;
	auto takeW = [&]( uint32_t words) { uint32_t at = w; w += (words + 3u) & ~3u; return at; };
	const uint32_t spillRules = maxRules - R;
	S.maxRules = maxRules;
	S.oCold = takeW( 8*maxRules);
	S.oHot = takeW( spillRules); S.oLink = takeW( 3*spillRules);
	S.oFree = takeW( spillRules);
	S.oEnt = takeW( 2*16*FAST_SPILL_BUCKET);
	S.maxStaged = maxStaged;
	S.oStaged = takeW( 8*maxStaged);
	S.oList = takeW( maxRules > FAST_LISTCAP ? maxRules : FAST_LISTCAP);
	S.oExp = takeW( (1u << expShift) * maxRules);
	S.totalWords = (w + 63u) & ~63u;
}

} // namespace
