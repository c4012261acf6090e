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
	if (ft.nofStopWords > 128) return "more than 128 stop words";
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

// LDS and spill layouts for the capacities (R rules, T bucket entries in LDS; maxRules in all)
void layoutFast( FastLdsLayout& L, FastSpillLayout& S, uint32_t R, uint32_t T, uint32_t nofStopWords, uint32_t maxRules, uint32_t maxStaged)
{
	T = (T + FAST_CHUNK-1) / FAST_CHUNK * FAST_CHUNK;
	if (T > FAST_MAXCHUNKS*FAST_CHUNK) T = FAST_MAXCHUNKS*FAST_CHUNK;
	if (R > maxRules) R = maxRules;
	uint32_t o = 0;
	auto take = [&]( uint32_t bytes) { uint32_t at = o; o += (bytes + 15u) & ~15u; return at; };
	L.R = R; L.T = T;
	L.oScalars = take( 32*4);
	L.oBSize = take( 16*4);
	L.oBChunks = take( 16*4);
	L.oWin = take( 64*2);
	L.oHot = take( R*4);
	L.oLink = take( 3*R*2);
	L.oNext = take( R*2);
	L.oFree = take( R*2);
	L.oEv = take( T*4);
	L.oTs = take( T*4);
	L.oChunkTab = take( 16*FAST_BUCKET_CHUNKS);
	L.oChunkFree = take( FAST_MAXCHUNKS);
	L.oStop = take( (nofStopWords ? nofStopWords : 1)*12);
	L.oList = take( FAST_LISTCAP*2);
	L.totalBytes = o;

	uint32_t w = 0;
	auto takeW = [&]( uint32_t words) { uint32_t at = w; w += (words + 3u) & ~3u; return at; };
	const uint32_t spillRules = maxRules - R;
	S.maxRules = maxRules;
	S.oCold = takeW( 8*maxRules);
	S.oHot = takeW( spillRules); S.oLink = takeW( 3*spillRules); S.oNext = takeW( spillRules);
	S.oFree = takeW( spillRules);
	S.oEnt = takeW( 2*(FAST_MAXCHUNKS*FAST_CHUNK - T));
	S.maxStaged = maxStaged;
	S.oStaged = takeW( 8*maxStaged);
	S.oList = takeW( maxRules);
	S.totalWords = (w + 63u) & ~63u;
}

} // namespace
