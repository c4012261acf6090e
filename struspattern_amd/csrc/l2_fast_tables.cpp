// Host side of the fast tier of level 2: decides whether a compiled rule set is FLAT (l2_fast.h) and
// builds the one-line-per-install table the fast kernel reads.
#include "l2_fast.h"
#include "l2_compile.hpp"
#include <map>
#include <set>
#include <cstring>

namespace spa {

static_assert( sizeof(FastKeyInst) == 64, "one install record per 64-byte line");
static_assert( sizeof(FastStatic) == 32, "compact static line");
static_assert( sizeof(FastKeyEntry) == sizeof(DevKeyEntry), "the fast kernel reads the general key table in place");

// Why a rule set is not eligible (diagnostics, SPA_L2_VERBOSE), empty when it is.
std::string buildFastTables( const FlatTables& ft, std::vector<FastKeyInst>& out, std::vector<FastStatic>* statics)
{
	out.clear();
	if (statics) statics->clear();
	if (ft.nofStopWords > FAST_MAXSTOP) return "more than 64 stop words";
	if (ft.keylist.size() >= (1u<<20)) return "more than 2^20 key list entries";	// (a rule keeps its install line in 20 bits)
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
	if (statics) { statics->resize( ft.keylist.size() ? ft.keylist.size() : 1); std::memset( statics->data(), 0, statics->size()*sizeof(FastStatic)); }
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
			ki.pastEvent = ref.pastEvent;
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
		// ---- static install (l2_fast.h): the key fires of every line simulated here, the ranks of what a 64-lane batch
		// appends counted here.  A batch with an alternative-keyed program (its replay depends on the document) or with a
		// program whose key fires capture a fourth item stays dynamic.
		for (uint32_t b0=0; b0<e.listCount; b0+=64)
		{
			const uint32_t nb = (e.listCount - b0) < 64u ? (e.listCount - b0) : 64u;
			FastKeyInst* B = &out[ e.listBegin + b0];
			bool isStatic = true;
			uint32_t bucketCnt[ 16], rangeCnt[ 64], rangeLast[ 64];
			std::memset( bucketCnt, 0, sizeof(bucketCnt)); std::memset( rangeCnt, 0, sizeof(rangeCnt)); std::memset( rangeLast, 0xFF, sizeof(rangeLast));
			int bucketLastLane[ 16], bucketLastSlot[ 16];
			for (int h=0; h<16; ++h) { bucketLastLane[ h] = -1; bucketLastSlot[ h] = -1; }
			uint32_t trigTotal = 0, firesTotal = 0, resTotal = 0, dispTotal = 0, itemsTotal = 0;
			for (uint32_t l=0; l<nb; ++l)
			{
				FastKeyInst& ki = B[ l];
				ki.hw0 = 0; ki.ranksA = 0; ki.ranksB = 0; ki.totals = 0; ki.items = 0; ki.flags = 0;
				if (ki.pastEvent) { isStatic = false; continue; }
				Sim sim;
				std::memset( &sim, 0, sizeof(sim));
				sim.value = ki.meta & FKI_VALUE_MASK; sim.count = (ki.meta >> FKI_COUNT_SHIFT) & FKI_COUNT_MASK;
				const uint32_t S = 1000;	// any position but 0 (position 0 takes the dynamic path: a start at position 0 counts as unset, cpp:919)
				for (int j=0; j<3; ++j) if (ki.trig[ j].info & FTI_KEY) fireLocal( sim, ki.trig[ j].info, S, 0/*key lexem*/, 1);
				if (sim.flags & S_ODD) { isStatic = false; continue; }
				uint32_t tmask = 0;
				for (int j=0; j<3; ++j)
				{
					if (!(ki.trig[ j].info & FTI_INSTALL)) continue;
					const uint32_t h = (ki.trig[ j].info >> FTI_BUCKET_SHIFT) & 15u;
					ki.ranksB |= (bucketCnt[ h] & 0xFFu) << (8*j);
					bucketCnt[ h] += 1; bucketLastLane[ h] = (int)l; bucketLastSlot[ h] = j;
					tmask |= 1u << j; trigTotal += 1;
				}
				ki.hw0 = sim.value | (sim.count << H_COUNT_SHIFT) | (uint32_t)H_ACTIVE | (tmask << H_TMASK_SHIFT) | (sim.nItems << H_NITEMS_SHIFT)
					| ((sim.flags & S_DONE) ? (uint32_t)H_DONE : 0u) | ((sim.flags & S_HASSTART) ? (uint32_t)H_HASSTART : 0u) | (ki.resultHandle ? (uint32_t)H_VISIBLE : 0u);
				if (sim.end) ki.flags |= FKF_END_SET;
				if (sim.flags & S_TOOK) ki.flags |= FKF_START_SET;
				const bool resultNow = (sim.flags & S_RESULT) && ki.resultHandle != 0;
				const bool disposeNow = (sim.flags & (S_DEL | S_FIN)) != 0;
				const uint32_t range = (ki.meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK;
				ki.ranksA = (rangeCnt[ range] & 0xFFu) | ((resTotal & 0xFFu) << 16) | ((dispTotal & 0xFFu) << 24);
				rangeCnt[ range] += 1; rangeLast[ range] = l;
				if (resultNow) { ki.flags |= FKF_RESULT_NOW; resTotal += 1; itemsTotal += sim.nItems; }
				if (disposeNow) { ki.flags |= FKF_DISPOSE_NOW; dispTotal += 1; ki.hw0 |= (uint32_t)H_LISTED; }	// (listed: the expiry row that meets the rule in the dispose list's batch leaves it alone)
				ki.items = (sim.it0 >> 24) | ((sim.it1 >> 24) << 8) | ((sim.it2 >> 24) << 16);
				firesTotal += sim.nFires;
			}
			if (!isStatic || trigTotal > 255u || firesTotal > 255u || itemsTotal > 255u) continue;
			for (int h=0; h<16; ++h) if (bucketLastLane[ h] >= 0) B[ bucketLastLane[ h]].ranksB |= 1u << (24 + bucketLastSlot[ h]);
			for (uint32_t rg=0; rg<64; ++rg) if (rangeCnt[ rg]) B[ rangeLast[ rg]].ranksA |= (rangeCnt[ rg] & 0xFFu) << 8;
			for (uint32_t l=0; l<nb; ++l)
			{
				B[ l].flags |= FKF_BATCH_STATIC;
				B[ l].totals = trigTotal | (firesTotal << 8) | (resTotal << 16) | (dispTotal << 24);
				B[ l].items |= itemsTotal << 24;
			}
			// compact form of the batch (l2_fast.h): every program installs at most two triggers
			if (!statics) continue;
			bool compact = true;
			for (uint32_t l=0; l<nb && compact; ++l)
			{
				int n = 0;
				for (int j=0; j<3; ++j) if (B[ l].trig[ j].info & FTI_INSTALL) ++n;
				if (n > 2) compact = false;
			}
			if (!compact) continue;
			FastStatic* C = &(*statics)[ e.listBegin + b0];
			for (uint32_t l=0; l<nb; ++l)
			{
				const FastKeyInst& ki = B[ l];
				FastStatic& fs = C[ l];
				int n = 0;
				for (int j=0; j<3; ++j)
				{
					const uint32_t info = ki.trig[ j].info;
					if (!(info & FTI_INSTALL)) continue;
					const uint32_t sv = (info & FTI_SIGVAL_MASK) | (((info >> FTI_SIGTYPE_SHIFT) & FTI_SIGTYPE_MASK) << 4) | ((info & FTI_HASVAR) ? 0x80u : 0u);
					const uint32_t ci = sv | ((info >> FTI_VAR_SHIFT) << 8) | (((info >> FTI_BUCKET_SHIFT) & 15u) << FSI_BUCKET_SHIFT)
						| (((ki.ranksB >> (8*j)) & 0xFFu) << FSI_RANK_SHIFT) | (((ki.ranksB >> (24+j)) & 1u) ? (uint32_t)FSI_LAST : 0u) | (uint32_t)FSI_PRESENT | ((uint32_t)j << FSI_SLOT_SHIFT);
					if (n == 0) { fs.ev0 = ki.trig[ j].event; fs.info0 = ci; } else { fs.ev1 = ki.trig[ j].event; fs.info1 = ci; }
					++n;
				}
				fs.hw0 = ki.hw0;
				fs.misc = ((ki.meta >> FKI_RANGE_SHIFT) & FKI_RANGE_MASK) | ((ki.ranksA & 0xFFu) << FSM_EXPRANK_SHIFT) | (((ki.ranksA >> 8) & 0xFFu) << FSM_EXPCLOSE_SHIFT)
					| ((ki.flags & FKF_END_SET) ? (uint32_t)FSM_END_SET : 0u) | ((ki.flags & FKF_START_SET) ? (uint32_t)FSM_START_SET : 0u)
					| ((ki.flags & FKF_RESULT_NOW) ? (uint32_t)FSM_RESULT_NOW : 0u) | ((ki.flags & FKF_DISPOSE_NOW) ? (uint32_t)FSM_DISPOSE_NOW : 0u) | (uint32_t)FSM_BATCH_COMPACT;
				fs.ranks = ((ki.ranksA >> 16) & 0xFFu) | (((ki.ranksA >> 24) & 0xFFu) << 8) | (itemsTotal << 16);
				fs.totals = ki.totals;
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
	S.oKi = takeW( spillRules); S.oLex0 = takeW( spillRules);
	S.oFree = takeW( spillRules);
	S.oEnt = takeW( 2*16*FAST_SPILL_BUCKET);
	S.maxStaged = maxStaged;
	S.oStaged = takeW( 8*maxStaged);
	S.oList = takeW( maxRules > FAST_LISTCAP ? maxRules : FAST_LISTCAP);
	S.oExp = takeW( (1u << expShift) * maxRules);
	S.totalWords = (w + 63u) & ~63u;
}

} // namespace
