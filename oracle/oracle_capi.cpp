// TEST INFRASTRUCTURE ONLY -- flat C entry points over the CPU oracle so that tests/ and
// bench.py's cpu_baseline leg can drive it through ctypes.  Not linked into the product.
#include "l2_oracle.hpp"
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>
#include <set>

using namespace oracle;

namespace {
struct L2Handle
{
	MatcherInstance inst;
	std::string lasterror;
};
template <class FN>
int guarded( L2Handle* h, FN fn)
{
	try { fn(); return 0; }
	catch (const std::exception& e) { h->lasterror = e.what(); return -1; }
}
}

extern "C" {

void* orc_l2_new() { return new L2Handle(); }
void orc_l2_free( void* h) { delete (L2Handle*)h; }
const char* orc_l2_last_error( void* h) { return ((L2Handle*)h)->lasterror.c_str(); }

int orc_l2_define_term_frequency( void* hp, uint32_t termid, double df)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.defineTermFrequency( termid, df); }); }
int orc_l2_push_term( void* hp, uint32_t termid)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.pushTerm( termid); }); }
int orc_l2_push_expression( void* hp, int op, uint32_t argc, uint32_t range, uint32_t cardinality)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.pushExpression( (JoinOp)op, argc, range, cardinality); }); }
int orc_l2_push_pattern( void* hp, const char* name)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.pushPattern( name); }); }
int orc_l2_attach_variable( void* hp, const char* name)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.attachVariable( name); }); }
int orc_l2_define_pattern( void* hp, const char* name, const char* fmt, int visible)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.definePattern( name, fmt?fmt:"", visible!=0); }); }
int orc_l2_define_option( void* hp, const char* name, double v)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.defineOption( name, v); }); }
int orc_l2_compile( void* hp)
{ L2Handle* h=(L2Handle*)hp; return guarded( h, [&]{ h->inst.compile(); }); }
uint32_t orc_l2_pattern_id( void* hp, const char* name) { return ((L2Handle*)hp)->inst.patternMap().get( name); }
uint32_t orc_l2_variable_id( void* hp, const char* name) { return ((L2Handle*)hp)->inst.variableMap().get( name); }

// Canonical serialisation of the compiled table (same format as sp_matcher_dump_table in the
// product, so tests can compare the two compilers word by word):
//   nPrograms, nKeyEvents, nStopWords,
//   per program 1..n: initsigval, initcount, event, resultHandle, formatHandle, positionRange, ntrig,
//                     ntrig x (event, isKey, sigtype, sigval, variable)        [list iteration order]
//   per key event (ascending id): event, n, n x (programidx, past_eventid)   [list iteration order]
//   stop words ascending
uint64_t orc_l2_dump_table( void* hp, uint32_t** out)
{
	const ProgramTable& pt = ((L2Handle*)hp)->inst.programTable();
	std::vector<uint32_t> buf;
	std::vector<uint32_t> keys = pt.keyEvents();
	std::sort( keys.begin(), keys.end());
	buf.push_back( (uint32_t)pt.nofPrograms());
	buf.push_back( (uint32_t)keys.size());
	buf.push_back( (uint32_t)pt.stopWords().size());
	for (uint32_t pi=1; pi<=pt.nofPrograms(); ++pi)
	{
		const Program& p = pt.program( pi);
		buf.push_back( p.slotDef.initsigval); buf.push_back( p.slotDef.initcount); buf.push_back( p.slotDef.event);
		buf.push_back( p.slotDef.resultHandle); buf.push_back( p.slotDef.formatHandle); buf.push_back( p.positionRange);
		std::size_t npos = buf.size(); buf.push_back( 0);
		uint32_t lst = p.triggerListIdx; const TriggerDef* td; uint32_t n=0;
		while (0!=(td=pt.triggerList().nextptr( lst)))
		{
			buf.push_back( td->event); buf.push_back( td->isKeyEvent); buf.push_back( td->sigtype);
			buf.push_back( td->sigval); buf.push_back( td->variable); ++n;
		}
		buf[ npos] = n;
	}
	for (std::size_t ki=0; ki<keys.size(); ++ki)
	{
		buf.push_back( keys[ki]);
		std::size_t npos = buf.size(); buf.push_back( 0);
		uint32_t lst = pt.getEventProgramList( keys[ki]); const ProgramTrigger* p; uint32_t n=0;
		while (0!=(p=pt.nextProgramPtr( lst))) { buf.push_back( p->programidx); buf.push_back( p->past_eventid); ++n; }
		buf[ npos] = n;
	}
	for (std::set<uint32_t>::const_iterator si=pt.stopWords().begin(); si!=pt.stopWords().end(); ++si) buf.push_back( *si);
	*out = (uint32_t*)std::malloc( buf.size()*sizeof(uint32_t)+4);
	std::memcpy( *out, buf.data(), buf.size()*sizeof(uint32_t));
	return buf.size();
}

struct OrcL2Out
{
	uint64_t nresults;
	uint64_t nitems;
	uint32_t* results;	// nresults x 9: handle, sord, eord, sseg, spos, eseg, epos, item_begin, item_count
	uint32_t* items;	// nitems x 7: variable, sord, eord, sseg, spos, eseg, epos
	uint64_t* doc_result_offsets; // ndocs+1
	uint64_t* doc_stats;	// ndocs x 4: programsInstalled, altKeyProgramsInstalled, signalsFired, sum of open triggers
	int32_t* doc_status;	// ndocs: 0 ok, -1 error
	uint32_t* result_format;// nresults (NULL when the matcher has no format strings)
	uint32_t* item_format;	// nitems x {format handle, nsub}: the nsub records after an item with a format are its arguments
};

namespace {
uint64_t countItems( const std::vector<ResultItem>& list)
{
	uint64_t n = list.size();
	for (std::size_t i=0; i<list.size(); ++i) n += countItems( list[i].args);
	return n;
}
void writeItems( const std::vector<ResultItem>& list, OrcL2Out* out, uint64_t& ip)
{
	for (std::size_t ii=0; ii<list.size(); ++ii)
	{
		const ResultItem& it = list[ii];
		const uint64_t at = ip++;
		uint32_t* q = out->items + at*7;
		q[0]=it.variable; q[1]=it.start_ordpos; q[2]=it.end_ordpos; q[3]=it.start_origseg; q[4]=it.start_origpos;
		q[5]=it.end_origseg; q[6]=it.end_origpos;
		writeItems( it.args, out, ip);
		if (out->item_format) { out->item_format[ 2*at] = it.formatHandle; out->item_format[ 2*at+1] = (uint32_t)(ip - at - 1); }
	}
}
}

// lexems: n x 5 u32 (id, ordpos, origseg, origpos, origsize); doc_offsets: ndocs+1 lexem indices.
// One MatcherContext per document (testRandomTokenPatternMatch.cpp:130-146), documents spread over
// nthreads threads sharing the immutable instance (same file :325-345).
int orc_l2_run_docs( void* hp, const uint32_t* lexems, const uint64_t* doc_offsets, uint64_t ndocs, int nthreads, OrcL2Out* out)
{
	L2Handle* h = (L2Handle*)hp;
	std::vector<std::vector<MatchResult> > perdoc( ndocs);
	out->doc_stats = (uint64_t*)std::calloc( ndocs*4+1, sizeof(uint64_t));
	out->doc_status = (int32_t*)std::calloc( ndocs+1, sizeof(int32_t));
	std::atomic<uint64_t> cursor(0);
	std::string firsterr;
	std::atomic<int> haserr(0);
	auto worker = [&]()
	{
		for (;;)
		{
			uint64_t di = cursor.fetch_add( 1);
			if (di >= ndocs) break;
			try
			{
				MatcherContext ctx( &h->inst);
				for (uint64_t li=doc_offsets[di]; li<doc_offsets[di+1]; ++li)
				{
					const uint32_t* lp = lexems + li*5;
					Lexem lx; lx.id=lp[0]; lx.ordpos=lp[1]; lx.origseg=lp[2]; lx.origpos=lp[3]; lx.origsize=lp[4];
					ctx.putInput( lx);
				}
				perdoc[ di] = ctx.fetchResults();
				uint64_t* st = out->doc_stats + di*4;
				st[0] = ctx.nofProgramsInstalled(); st[1] = ctx.nofAltKeyProgramsInstalled();
				st[2] = ctx.nofSignalsFired(); st[3] = (uint64_t)ctx.nofOpenPatterns();
			}
			catch (const std::exception& e)
			{
				out->doc_status[ di] = -1;
				if (!haserr.exchange( 1)) firsterr = e.what();
			}
		}
	};
	if (nthreads <= 1) worker();
	else
	{
		std::vector<std::thread> th;
		for (int ti=0; ti<nthreads; ++ti) th.push_back( std::thread( worker));
		for (std::size_t ti=0; ti<th.size(); ++ti) th[ti].join();
	}
	uint64_t nres=0, nitems=0;
	for (uint64_t di=0; di<ndocs; ++di)
	{
		nres += perdoc[di].size();
		for (std::size_t ri=0; ri<perdoc[di].size(); ++ri) nitems += countItems( perdoc[di][ri].items);
	}
	out->nresults = nres; out->nitems = nitems;
	out->results = (uint32_t*)std::malloc( (nres*9+1)*sizeof(uint32_t));
	out->items = (uint32_t*)std::malloc( (nitems*7+1)*sizeof(uint32_t));
	out->doc_result_offsets = (uint64_t*)std::malloc( (ndocs+1)*sizeof(uint64_t));
	out->result_format = 0; out->item_format = 0;
	if (h->inst.nofFormats())
	{
		out->result_format = (uint32_t*)std::malloc( (nres+1)*sizeof(uint32_t));
		out->item_format = (uint32_t*)std::malloc( (nitems+1)*2*sizeof(uint32_t));
	}
	uint64_t rp=0, ip=0;
	for (uint64_t di=0; di<ndocs; ++di)
	{
		out->doc_result_offsets[ di] = rp;
		for (std::size_t ri=0; ri<perdoc[di].size(); ++ri,++rp)
		{
			const MatchResult& m = perdoc[di][ri];
			uint32_t* r = out->results + rp*9;
			r[0]=m.resultHandle; r[1]=m.start_ordpos; r[2]=m.end_ordpos; r[3]=m.start_origseg; r[4]=m.start_origpos;
			r[5]=m.end_origseg; r[6]=m.end_origpos; r[7]=(uint32_t)ip; r[8]=(uint32_t)countItems( m.items);
			if (out->result_format) out->result_format[ rp] = m.formatHandle;
			writeItems( m.items, out, ip);
		}
	}
	out->doc_result_offsets[ ndocs] = rp;
	if (haserr.load()) { h->lasterror = firsterr; return -1; }
	return 0;
}

void orc_l2_free_out( OrcL2Out* out)
{
	std::free( out->results); std::free( out->items); std::free( out->doc_result_offsets);
	std::free( out->doc_stats); std::free( out->doc_status); std::free( out->result_format); std::free( out->item_format);
	std::memset( out, 0, sizeof(*out));
}

void orc_free( void* p) { std::free( p); }

} // extern "C"

// ------------------------------------------------------------------ level 1
#include "l1_oracle.hpp"
namespace {
struct L1Handle { oracle::LexerInstance inst; std::string lasterror; };
template <class FN>
int guarded1( L1Handle* h, FN fn)
{
	try { fn(); return 0; }
	catch (const std::exception& e) { h->lasterror = e.what(); return -1; }
}
}
extern "C" {
void* orc_l1_new() { return new L1Handle(); }
void orc_l1_free( void* h) { delete (L1Handle*)h; }
const char* orc_l1_last_error( void* h) { return ((L1Handle*)h)->lasterror.c_str(); }
int orc_l1_define_lexem( void* hp, uint32_t id, const char* expr, uint32_t resultIndex, uint32_t level, int posbind)
{ L1Handle* h=(L1Handle*)hp; return guarded1( h, [&]{ h->inst.defineLexem( id, expr, resultIndex, level, (oracle::PosBind)posbind); }); }
int orc_l1_define_symbol( void* hp, uint32_t symbolid, uint32_t patternid, const char* name)
{ L1Handle* h=(L1Handle*)hp; return guarded1( h, [&]{ h->inst.defineSymbol( symbolid, patternid, name); }); }
int orc_l1_define_option( void* hp, const char* name, double v)
{ L1Handle* h=(L1Handle*)hp; return guarded1( h, [&]{ h->inst.defineOption( name, v); }); }
int orc_l1_compile( void* hp)
{ L1Handle* h=(L1Handle*)hp; return guarded1( h, [&]{ h->inst.compile(); }); }
uint32_t orc_l1_get_symbol( void* hp, uint32_t patternid, const char* name) { return ((L1Handle*)hp)->inst.getSymbol( patternid, name); }

// documents: concatenated bytes + ndocs+1 offsets.  out_lexems: malloc'd n x 4 u32 (id, ordpos, origpos, origsize);
// out_doc_offsets: malloc'd ndocs+1 u64.  raw != 0: returns the raw report stream (idx, from, to, 0) instead.
int orc_l1_match_docs( void* hp, const char* text, const uint64_t* doc_offsets, uint64_t ndocs, int nthreads, int raw,
			uint32_t** out_lexems, uint64_t** out_doc_offsets)
{
	L1Handle* h = (L1Handle*)hp;
	std::vector<std::vector<uint32_t> > perdoc( ndocs);
	std::atomic<uint64_t> cursor(0);
	std::atomic<int> haserr(0);
	std::string firsterr;
	auto worker = [&]()
	{
		for (;;)
		{
			uint64_t di = cursor.fetch_add( 1);
			if (di >= ndocs) break;
			try
			{
				const char* src = text + doc_offsets[ di];
				size_t len = (size_t)(doc_offsets[ di+1] - doc_offsets[ di]);
				std::vector<uint32_t>& o = perdoc[ di];
				if (raw)
				{
					std::vector<oracle::RawMatch> r = h->inst.rawMatches( src, len);
					for (size_t i=0; i<r.size(); ++i) { o.push_back( r[i].idx); o.push_back( r[i].from); o.push_back( r[i].to); o.push_back( 0); }
				}
				else
				{
					std::vector<oracle::LexemOut> r = h->inst.match( src, len);
					for (size_t i=0; i<r.size(); ++i) { o.push_back( r[i].id); o.push_back( r[i].ordpos); o.push_back( r[i].origpos); o.push_back( r[i].origsize); }
				}
			}
			catch (const std::exception& e)
			{
				if (!haserr.exchange( 1)) firsterr = e.what();
			}
		}
	};
	if (nthreads <= 1) worker();
	else
	{
		std::vector<std::thread> th;
		for (int ti=0; ti<nthreads; ++ti) th.push_back( std::thread( worker));
		for (size_t ti=0; ti<th.size(); ++ti) th[ti].join();
	}
	uint64_t total = 0;
	for (uint64_t di=0; di<ndocs; ++di) total += perdoc[ di].size()/4;
	*out_lexems = (uint32_t*)std::malloc( (total*4+1)*sizeof(uint32_t));
	*out_doc_offsets = (uint64_t*)std::malloc( (ndocs+1)*sizeof(uint64_t));
	uint64_t p = 0;
	for (uint64_t di=0; di<ndocs; ++di)
	{
		(*out_doc_offsets)[ di] = p;
		if (!perdoc[ di].empty()) std::memcpy( *out_lexems + p*4, perdoc[ di].data(), perdoc[ di].size()*sizeof(uint32_t));
		p += perdoc[ di].size()/4;
	}
	(*out_doc_offsets)[ ndocs] = p;
	if (haserr.load()) { h->lasterror = firsterr; return -1; }
	return 0;
}
} // extern "C"
