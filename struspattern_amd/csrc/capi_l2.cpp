// C-ABI of level 2 (include/strus_pattern_amd.h): rule compiler handle + GPU match context.
// No CPU fallback: a context cannot be created without a usable HIP device.
#include "../../include/strus_pattern_amd.h"
#include "l2_compile.hpp"
#include "l2_device.h"
#include "l2_fast.h"
#include "l2_join.h"
#include "hip_util.hpp"
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstdio>
#include <unistd.h>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace spa {
hipError_t launchL2Match( const L2Params& P, unsigned nblocks, hipStream_t stream);
hipError_t launchL2Fast( const FastParams& P, unsigned variant, unsigned nblocks, hipStream_t stream);
int fastBlocksPerCU( unsigned variant);
void fastCapacities( unsigned variant, uint32_t& R, uint32_t& T);
std::string buildFastTables( const FlatTables& ft, std::vector<FastKeyInst>& out, std::vector<FastStatic>* statics);
std::string buildJoinTables( const FlatTables& ft, std::vector<JoinKey>& keytab, std::vector<JoinRule>& rules, std::vector<uint32_t>& filter, uint32_t& maxRange, uint32_t& delimiter);
hipError_t launchL2Join( const JoinParams& P, unsigned nwaves, hipStream_t stream);
void layoutFast( FastSpillLayout& S, uint32_t bucketMeta[16], uint32_t& expShift, const std::vector<FastKeyInst>& keyinst, uint32_t R, uint32_t T, uint32_t maxRules, uint32_t maxStaged);
}

using namespace spa;

#ifndef SPA_L2_WAVES_PER_CU
#define SPA_L2_WAVES_PER_CU 12
#endif

struct sp_matcher
{
	RuleCompiler compiler;
	bool compiled;		// compile() is optional for the matcher (reference: tests/randomTokenPatternMatch :317-320)
	mutable std::string lasterror;
	sp_matcher() :compiled(false){}
};

namespace {

template <class FN>
int guardedCall( std::string& err, int errcode, FN fn)
{
	try { fn(); return SP_OK; }
	catch (const std::bad_alloc&) { err = "memory allocation error in strus pattern"; return SP_ERR_NOMEM; }
	catch (const HipError& e) { err = e.what(); return SP_ERR_DEVICE; }
	catch (const std::exception& e) { err = e.what(); return errcode; }
}

uint32_t alignUp( uint32_t v, uint32_t a) { return (v + a-1) / a * a; }

void layoutArena( ArenaLayout& L)
{
	uint32_t o = 0;
	L.oRules = o;	o += L.maxRules*32;			// 128-byte blocks: rule + its 4 trigger slots
	L.oTrigs = 0; L.oBIdx = 0; L.oTrigFree = 0;		// (unused: triggers live in the rule blocks)
	L.oBEvent = o;	o += alignUp( 2*16*L.bucketCap, 32);	// {event, trigger id} pairs
	L.oBSize = o;	o += 16;
	L.oWindow = o;	o += 64;
	L.oHeap = o;	o += alignUp( L.maxHeap*2, 4);
	L.oFollow = o;	o += alignUp( L.maxFollow*12, 4);
	L.oDispose = o;	o += alignUp( L.maxDispose, 4);
	L.oStop = o;	o += alignUp( (L.nStop?L.nStop:1)*12, 4);
	L.oItems = o;	o += alignUp( L.maxItems*12, 4);
	L.oRefs = o;	o += alignUp( L.maxRefs*2, 4);
	L.oGStack = o;	o += alignUp( L.maxGStack, 4);
	L.oStaged = o;	o += alignUp( L.maxStaged*8, 4);
	L.oRuleFree = o; o += alignUp( L.maxRules, 4);
	L.oItemFree = o; o += alignUp( L.maxItems, 4);
	L.oRefFree = o;	o += alignUp( L.maxRefs, 4);
	// expiry window: lists of up to 8 chunks per position; the pool covers every live rule plus one
	// partly filled chunk per position
	L.winChunk = L.winCap/8 < 16 ? 16 : L.winCap/8;
	L.winChunks = L.maxRules/L.winChunk + 64;
	L.oWinArr = o;	o += alignUp( L.winChunks*L.winChunk, 4);
	L.oWinChunk = o; o += 64*8;
	L.oWinFree = o;	o += alignUp( L.winChunks, 4);
	L.oScratch = o;	o += alignUp( 16*L.scratchCap, 4);
	L.totalWords = alignUp( o, 64);
}

} // namespace

struct sp_matcher_ctx
{
	const sp_matcher* inst;
	int device;
	std::string lasterror;
	// device tables
	DeviceBuffer dPrograms, dTrigdefs, dKeytab, dKeylist;
	uint32_t keymask, nofStopWords;
	// fast tier (l2_fast.h): flat rule sets run with their hot state in LDS; the general kernel takes what it hands over
	bool fast;
	std::string whyNotFast;
	DeviceBuffer dKeyinst, dStatics, dSpill, dFallbackList;
	// join prototype (l2_join.h, opt-in by SPA_L2_JOIN=1): result sets without materialised rule instances
	bool join; std::string whyNotJoin; uint32_t joinKeymask, joinMaxRange, joinDelimiter; DeviceBuffer dJoinKeytab, dJoinRules, dJoinFilter, dJoinCounts;
	std::vector<FastKeyInst> fastKeyinst;
	FastSpillLayout fastSpill; uint32_t fastBucketMeta[ 16]; uint32_t fastExpShift;
	unsigned fastWaves, fastBlocksPerCU, fastVariant;	// variant: kernel instance = LDS capacities (l2_fast_kernel.hip)
	uint32_t fastMaxRules, fastMaxStaged;
	// working memory
	ArenaLayout arena;
	DeviceBuffer dArena; unsigned arenaWaves;
	DeviceBuffer dCursor, dCounters;
	// batch buffers (grown on demand)
	DeviceBuffer dLexems, dOrigseg, dDocOffsets, dResults, dItems, dDocRange, dDocStats, dDocStatus;
	DeviceBuffer dResultFormat, dItemFormat;	// only for matchers with format strings
	bool withFormats;
	std::vector<uint32_t> curResultFormat, curItemFormat;	// of the last sp_matcher_ctx_fetch_results
	uint64_t resultCapacity, itemCapacity, minResultCapacity, minItemCapacity;
	size_t lastNdocs;
	hipEvent_t evStart, evStop; bool evValid;
	hipStream_t lastStream;
	hipStream_t own;		// the context's own stream (non-blocking): see sp_lexer_ctx
	bool withItems;
	unsigned numCUs;
	// single-document mode
	std::vector<sp_lexem_t> curLexems;
	std::vector<uint32_t> curOrigseg; bool curHasSeg;
	sp_matcher_stats_t lastStats;

	sp_matcher_ctx() :inst(0),device(0),keymask(0),nofStopWords(0),fast(false),join(false),joinKeymask(0),joinMaxRange(0),joinDelimiter(0),fastWaves(0),fastBlocksPerCU(0),fastVariant(4),fastMaxRules(2048),fastMaxStaged(32768),arenaWaves(0),withFormats(false),resultCapacity(0),itemCapacity(0),minResultCapacity(0),minItemCapacity(0)
		,lastNdocs(0),evStart(0),evStop(0),evValid(false),lastStream(0),own(0),withItems(true),numCUs(256),curHasSeg(false)
	{
		std::memset( &arena, 0, sizeof(arena));
		std::memset( &fastSpill, 0, sizeof(fastSpill)); std::memset( fastBucketMeta, 0, sizeof(fastBucketMeta));
		std::memset( &lastStats, 0, sizeof(lastStats));
		// small defaults (a document's hot state should stay cache and TLB friendly); every capacity
		// doubles automatically when a document overflows it (SP_DOC_ERR_ARENA -> grow -> rerun)
		arena.maxRules = 1024; arena.maxTrigs = 1024; arena.bucketCap = 256; arena.maxItems = 2048;
		arena.maxRefs = 1024; arena.maxFollow = 256; arena.maxDispose = 512; arena.maxHeap = 256;
		arena.maxGStack = 64; arena.maxStaged = 1024; arena.winCap = 128; arena.scratchCap = 256;
	}
};

// a copy on the context's own stream, complete when the call returns
static void copySync( sp_matcher_ctx* c, void* dst, const void* src, size_t n, hipMemcpyKind kind)
{
	HIP_CHECK( hipMemcpyAsync( dst, src, n, kind, c->own));
	HIP_CHECK( hipStreamSynchronize( c->own));
}

extern "C" {

const char* sp_version(void) { return "struspattern_amd 0.1 (gfx950)"; }

int sp_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount( &n) != hipSuccess) return 0;
	return n;
}

void sp_free( void* p) { std::free( p); }

// test hook: device allocations of at least `bytes` fail like an out-of-memory hipMalloc (0 = off)
void sp_test_fail_alloc_above( uint64_t bytes) { allocFailureThreshold().store( bytes, std::memory_order_relaxed); }

// ------------------------------------------------------------------ instance
sp_matcher_t* sp_matcher_create(void)
{
	try { return new sp_matcher(); } catch (...) { return 0; }
}
void sp_matcher_free( sp_matcher_t* m) { delete m; }
const char* sp_matcher_last_error( const sp_matcher_t* m) { return m->lasterror.c_str(); }

#define MGUARD( CODE, BODY) return guardedCall( m->lasterror, CODE, [&]{ BODY; })

int sp_matcher_define_term_frequency( sp_matcher_t* m, uint32_t termid, double df)
{ MGUARD( SP_ERR_INVALID, m->compiler.defineTermFrequency( termid, df)); }
int sp_matcher_push_term( sp_matcher_t* m, uint32_t termid)
{ MGUARD( SP_ERR_INVALID, m->compiler.pushTerm( termid)); }
int sp_matcher_push_expression( sp_matcher_t* m, int joinop, size_t argc, uint32_t range, uint32_t cardinality)
{ MGUARD( SP_ERR_INVALID, m->compiler.pushExpression( joinop, argc, range, cardinality)); }
int sp_matcher_push_pattern( sp_matcher_t* m, const char* name)
{ MGUARD( SP_ERR_INVALID, m->compiler.pushPattern( name ? name : "")); }
int sp_matcher_attach_variable( sp_matcher_t* m, const char* name)
{ MGUARD( SP_ERR_INVALID, m->compiler.attachVariable( name ? name : "")); }
int sp_matcher_define_pattern( sp_matcher_t* m, const char* name, const char* formatstring, int visible)
{ MGUARD( SP_ERR_INVALID, m->compiler.definePattern( name ? name : "", formatstring ? formatstring : "", visible != 0)); }
int sp_matcher_define_option( sp_matcher_t* m, const char* name, double value)
{ MGUARD( SP_ERR_INVALID, m->compiler.defineOption( name ? name : "", value)); }
int sp_matcher_compile( sp_matcher_t* m)
{ MGUARD( SP_ERR_COMPILE, m->compiler.compile(); m->compiled = true); }

uint32_t sp_matcher_pattern_id( const sp_matcher_t* m, const char* name) { return m->compiler.patterns().get( name); }
const char* sp_matcher_pattern_name( const sp_matcher_t* m, uint32_t handle) { return m->compiler.patterns().key( handle); }
uint32_t sp_matcher_variable_id( const sp_matcher_t* m, const char* name) { return m->compiler.variables().get( name); }
const char* sp_matcher_variable_name( const sp_matcher_t* m, uint32_t variable) { return m->compiler.variables().key( variable); }

uint32_t sp_matcher_format_count( const sp_matcher_t* m) { return m->compiler.formatCount(); }
const char* sp_matcher_format_string( const sp_matcher_t* m, uint32_t format_handle) { return m->compiler.formatString( format_handle); }

// which kernel the context's batches run on: 0 = general, 1 = LDS-resident (flat rule sets), 2 = join prototype (SPA_L2_JOIN=1)
int sp_matcher_ctx_kernel_kind( const sp_matcher_ctx_t* c) { return c->join ? 2 : c->fast ? 1 : 0; }
// name of the kernel that does a batch's work (the instance of the LDS-resident kernel is picked by SPA_L2_FAST_SIZE, default n)
const char* sp_matcher_ctx_kernel_name( const sp_matcher_ctx_t* c)
{
	if (c->join) return "spa_l2_join_kernel";
	if (!c->fast) return "spa_l2_match_kernel";
	static const char* names[ 5] = {"spa_l2_fast_kernel_s", "spa_l2_fast_kernel_m", "spa_l2_fast_kernel_l", "spa_l2_fast_kernel_t", "spa_l2_fast_kernel_n"};
	return names[ c->fastVariant < 5 ? c->fastVariant : 4];
}

// 1 when the compiled rule set is flat (l2_fast.h) and runs on the LDS-resident kernel, else 0 with the reason
int sp_matcher_fast_tier( const sp_matcher_t* m, char* why, size_t whysize)
{
	try
	{
		FlatTables ft;
		m->compiler.flatten( ft);
		std::vector<FastKeyInst> ki;
		const std::string reason = buildFastTables( ft, ki, 0);
		if (why && whysize) { std::strncpy( why, reason.c_str(), whysize-1); why[ whysize-1] = 0; }
		return reason.empty() ? 1 : 0;
	}
	catch (const std::exception& e)
	{
		if (why && whysize) { std::strncpy( why, e.what(), whysize-1); why[ whysize-1] = 0; }
		return 0;
	}
}

// the rule set as a blob (tables, names, format strings, options) and back: SURVEY.md 8(f).4
int sp_matcher_serialize( const sp_matcher_t* m, void** blob, size_t* size)
{
	*blob = 0; *size = 0;
	return guardedCall( m->lasterror, SP_ERR_INVALID, [&]{
		std::vector<uint8_t> buf;
		m->compiler.save( buf, m->compiled);
		*blob = std::malloc( buf.size() ? buf.size() : 1);
		if (!*blob) throw std::bad_alloc();
		std::memcpy( *blob, buf.data(), buf.size());
		*size = buf.size();
	});
}
sp_matcher_t* sp_matcher_deserialize( const void* blob, size_t size, char* err, size_t errsize)
{
	sp_matcher* m = 0;
	try
	{
		m = new sp_matcher();
		m->compiled = m->compiler.load( blob, size);
		return m;
	}
	catch (const std::exception& e)
	{
		if (err && errsize) { std::strncpy( err, e.what(), errsize-1); err[ errsize-1] = 0; }
		delete m;
		return 0;
	}
}

size_t sp_matcher_dump_table( const sp_matcher_t* m, uint32_t** out)
{
	std::vector<uint32_t> buf = m->compiler.dump();
	*out = (uint32_t*)std::malloc( buf.size()*sizeof(uint32_t) + 4);
	if (!*out) return 0;
	std::memcpy( *out, buf.data(), buf.size()*sizeof(uint32_t));
	return buf.size();
}

// ------------------------------------------------------------------ context
sp_matcher_ctx_t* sp_matcher_ctx_create( const sp_matcher_t* m, int device)
{
	sp_matcher_ctx* c = 0;
	try
	{
		c = new sp_matcher_ctx();
		c->inst = m; c->device = device;
		int ndev = 0;
		hipError_t e = hipGetDeviceCount( &ndev);
		if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
		{
			m->lasterror = "no usable HIP device: the rule automaton runs on the GPU only (no CPU fallback)";
			delete c; return 0;
		}
		HIP_CHECK( hipSetDevice( device));
		hipDeviceProp_t prop;
		HIP_CHECK( hipGetDeviceProperties( &prop, device));
		c->numCUs = prop.multiProcessorCount > 0 ? (unsigned)prop.multiProcessorCount : 256u;

		FlatTables ft;
		m->compiler.flatten( ft);
		c->dPrograms.upload( ft.programs.data(), ft.programs.size()*sizeof(DevProgram));
		c->dTrigdefs.upload( ft.trigdefs.data(), ft.trigdefs.size()*sizeof(DevTrigDef));
		c->dKeytab.upload( ft.keytab.data(), ft.keytab.size()*sizeof(DevKeyEntry));
		c->dKeylist.upload( ft.keylist.data(), ft.keylist.size()*sizeof(DevKeyRef));
		c->keymask = (uint32_t)ft.keytab.size()-1;
		c->nofStopWords = ft.nofStopWords;
		c->arena.nStop = ft.nofStopWords;
		c->withFormats = m->compiler.formatCount() != 0;
		{
			// fast tier: eligible rule sets get the one-line-per-install table (SPA_L2_FAST=0 keeps everything on the general kernel)
			std::vector<FastKeyInst> ki;
			std::vector<FastStatic> ks;
			c->whyNotFast = buildFastTables( ft, ki, &ks);
			const char* sw = getenv( "SPA_L2_FAST");
			if (sw && sw[0] == '0') c->whyNotFast = "disabled by SPA_L2_FAST=0";
			c->fast = c->whyNotFast.empty();
			if (c->fast)
			{
				if (ki.empty()) ki.resize( 1);
				c->dKeyinst.upload( ki.data(), ki.size()*sizeof(FastKeyInst));
				c->dStatics.upload( ks.data(), ks.size()*sizeof(FastStatic));
				c->fastKeyinst.swap( ki);
				// SPA_L2_FAST_SIZE=s|m|l picks the kernel instance (LDS capacities; t = the tiny one of the tests); the spill area takes what does not fit
				if (const char* e = getenv( "SPA_L2_FAST_SIZE")) c->fastVariant = (e[0] == 's') ? 0u : (e[0] == 'm') ? 1u : (e[0] == 'l') ? 2u : (e[0] == 't') ? 3u : 4u;
				if (const char* e = getenv( "SPA_L2_FAST_MAXRULES")) c->fastMaxRules = (uint32_t)atoi( e);
				if (const char* e = getenv( "SPA_L2_FAST_MAXSTAGED")) c->fastMaxStaged = (uint32_t)atoi( e);
				if (c->fastMaxRules > 4095) c->fastMaxRules = 4095;		// trigger ids are 14 bits (rule << 2 | slot)
			}
			if (getenv( "SPA_L2_VERBOSE")) fprintf( stderr, "[spa] fast tier: %s\n", c->fast ? "on" : c->whyNotFast.c_str());
		}
		if (const char* e = getenv( "SPA_L2_JOIN"))
		{
			if (e[0] == '1')
			{
				std::vector<JoinKey> jk; std::vector<JoinRule> jr; std::vector<uint32_t> jf;
				c->whyNotJoin = buildJoinTables( ft, jk, jr, jf, c->joinMaxRange, c->joinDelimiter);
				c->join = c->whyNotJoin.empty();
				if (c->join)
				{
					c->dJoinKeytab.upload( jk.data(), jk.size()*sizeof(JoinKey)); c->dJoinRules.upload( jr.data(), jr.size()*sizeof(JoinRule)); c->dJoinFilter.upload( jf.data(), jf.size()*sizeof(uint32_t));
					c->joinKeymask = (uint32_t)jk.size()-1;
				}
				if (getenv( "SPA_L2_VERBOSE")) fprintf( stderr, "[spa] join prototype: %s\n", c->join ? "on" : c->whyNotJoin.c_str());
			}
		}
		c->dCursor.alloc( 256);		// u32: [0] fast cursor, [1] general cursor (list mode), [2] hand-over count, [16..31] hand-over reasons, [32..47] phase profile (u64 x 8)
		c->dCounters.alloc( SPC_COUNT*sizeof(uint64_t));
		HIP_CHECK( hipStreamCreateWithFlags( &c->own, hipStreamNonBlocking));
		HIP_CHECK( hipEventCreate( &c->evStart));
		HIP_CHECK( hipEventCreate( &c->evStop));
		return c;
	}
	catch (const std::exception& e)
	{
		m->lasterror = e.what();
		delete c;
		return 0;
	}
}

void sp_matcher_ctx_free( sp_matcher_ctx_t* c)
{
	if (!c) return;
	if (c->own) { (void)hipSetDevice( c->device); (void)hipStreamSynchronize( c->own); (void)hipStreamDestroy( c->own); }
	if (c->evStart) (void)hipEventDestroy( c->evStart);
	if (c->evStop) (void)hipEventDestroy( c->evStop);
	delete c;
}
const char* sp_matcher_ctx_last_error( const sp_matcher_ctx_t* c) { return c->lasterror.c_str(); }

int sp_matcher_ctx_set_arena( sp_matcher_ctx_t* c, uint32_t max_rules, uint32_t max_triggers, uint32_t bucket_capacity,
				uint32_t max_items, uint32_t max_follow)
{
	if (max_rules) { c->arena.maxRules = max_rules; c->arena.maxHeap = max_rules; c->arena.maxDispose = max_rules; c->arena.winCap = max_rules/4 < 64 ? 64 : max_rules/4; }
	if (max_triggers) c->arena.maxTrigs = max_triggers;
	if (bucket_capacity) c->arena.bucketCap = bucket_capacity;
	if (max_items) { c->arena.maxItems = max_items; c->arena.maxRefs = max_items; }
	if (max_follow) { c->arena.maxFollow = max_follow; }
	c->arenaWaves = 0;	// forces re-layout at the next launch
	return SP_OK;
}

} // extern "C"

namespace {

// Host copy of the device results of the last launch for the documents [firstDoc, firstDoc+ndocs), regrouped
// by document (the device appends whole documents in completion order), with the `exclusive`
// elimination of fetchResults applied on the way.  The whole batch is copied in bulk; a sub-range
// copies only the result and item blocks of its own documents.
void copyOutBatch( sp_matcher_ctx* c, size_t firstDoc, size_t ndocs, const uint64_t* counters, sp_match_batch_t* out)
{
	const bool whole = (firstDoc == 0 && ndocs == c->lastNdocs);
	std::vector<uint64_t> range( ndocs*2+2);
	if (ndocs) copySync( c, range.data(), (const uint64_t*)c->dDocRange.ptr + 2*firstDoc, ndocs*2*sizeof(uint64_t), hipMemcpyDeviceToHost);
	out->ndocs = ndocs;
	out->doc_stats = (uint64_t*)std::malloc( (ndocs*4+1)*sizeof(uint64_t));
	out->doc_status = (int32_t*)std::malloc( (ndocs+1)*sizeof(int32_t));
	out->doc_result_offsets = (uint64_t*)std::malloc( (ndocs+1)*sizeof(uint64_t));
	if (!out->doc_stats || !out->doc_status || !out->doc_result_offsets) throw std::bad_alloc();
	if (ndocs)
	{
		copySync( c, out->doc_stats, (const uint64_t*)c->dDocStats.ptr + 4*firstDoc, ndocs*4*sizeof(uint64_t), hipMemcpyDeviceToHost);
		copySync( c, out->doc_status, (const int32_t*)c->dDocStatus.ptr + firstDoc, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
	}
	const uint64_t devResults = counters[ SPC_RESULTS] < c->resultCapacity ? counters[ SPC_RESULTS] : c->resultCapacity;
	const uint64_t devItems = counters[ SPC_ITEMS] < c->itemCapacity ? counters[ SPC_ITEMS] : c->itemCapacity;
	std::vector<sp_result_t> raw;
	std::vector<sp_result_item_t> rawitems;
	std::vector<uint32_t> rawrf, rawif;
	if (whole)
	{
		raw.resize( devResults+1); rawitems.resize( devItems+1);
		if (devResults) copySync( c, raw.data(), c->dResults.ptr, devResults*sizeof(sp_result_t), hipMemcpyDeviceToHost);
		if (devItems) copySync( c, rawitems.data(), c->dItems.ptr, devItems*sizeof(sp_result_item_t), hipMemcpyDeviceToHost);
		if (c->withFormats)
		{
			rawrf.resize( devResults+1); rawif.resize( 2*devItems+2);
			if (devResults) copySync( c, rawrf.data(), c->dResultFormat.ptr, devResults*sizeof(uint32_t), hipMemcpyDeviceToHost);
			if (devItems) copySync( c, rawif.data(), c->dItemFormat.ptr, 2*devItems*sizeof(uint32_t), hipMemcpyDeviceToHost);
		}
	}
	else
	{
		// the blocks of the wanted documents only, packed one after the other; ranges and item indices are rebased
		uint64_t nres = 0;
		for (size_t di=0; di<ndocs; ++di) if (out->doc_status[ di] == 0 && range[ 2*di] + range[ 2*di+1] <= devResults) nres += range[ 2*di+1];
		raw.resize( nres+1);
		if (c->withFormats) rawrf.resize( nres+1);
		uint64_t rp0 = 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			const uint64_t b = range[ 2*di], n = range[ 2*di+1];
			if (out->doc_status[ di] != 0 || b + n > devResults) { range[ 2*di+1] = 0; continue; }
			if (n) copySync( c, raw.data() + rp0, (const sp_result_t*)c->dResults.ptr + b, n*sizeof(sp_result_t), hipMemcpyDeviceToHost);
			if (n && c->withFormats) copySync( c, rawrf.data() + rp0, (const uint32_t*)c->dResultFormat.ptr + b, n*sizeof(uint32_t), hipMemcpyDeviceToHost);
			range[ 2*di] = rp0; rp0 += n;
		}
		uint64_t nitems = 0;
		for (uint64_t ri=0; ri<nres; ++ri) nitems += raw[ ri].item_count;
		rawitems.resize( nitems+1);
		if (c->withFormats) rawif.resize( 2*nitems+2);
		uint64_t ip0 = 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			// the items of one document are one block, in result order
			const uint64_t b = range[ 2*di], n = range[ 2*di+1];
			uint64_t first = 0, cnt = 0;
			for (uint64_t ri=0; ri<n; ++ri) if (raw[ b+ri].item_count) { if (!cnt) first = raw[ b+ri].item_begin; cnt += raw[ b+ri].item_count; }
			if (!cnt) continue;
			if (first + cnt > devItems) throw std::runtime_error( "item block of a document lies outside the device buffer");
			copySync( c, rawitems.data() + ip0, (const sp_result_item_t*)c->dItems.ptr + first, cnt*sizeof(sp_result_item_t), hipMemcpyDeviceToHost);
			if (c->withFormats) copySync( c, rawif.data() + 2*ip0, (const uint32_t*)c->dItemFormat.ptr + 2*first, 2*cnt*sizeof(uint32_t), hipMemcpyDeviceToHost);
			for (uint64_t ri=0; ri<n; ++ri) if (raw[ b+ri].item_count) raw[ b+ri].item_begin = (uint32_t)(raw[ b+ri].item_begin - first + ip0);
			ip0 += cnt;
		}
	}
	// regroup by document (the device appends whole documents in completion order)
	uint64_t total = 0, totalItems = 0;
	for (size_t di=0; di<ndocs; ++di)
	{
		if (out->doc_status[ di] != 0) { range[ 2*di+1] = 0; continue; }
		total += range[ 2*di+1];
		for (uint64_t ri=0; ri<range[ 2*di+1]; ++ri) totalItems += raw[ range[ 2*di]+ri].item_count;
	}
	out->results = (sp_result_t*)std::malloc( (total+1)*sizeof(sp_result_t));
	out->items = (sp_result_item_t*)std::malloc( (totalItems+1)*sizeof(sp_result_item_t));
	if (!out->results || !out->items) throw std::bad_alloc();
	if (c->withFormats)
	{
		out->result_format = (uint32_t*)std::malloc( (total+1)*sizeof(uint32_t));
		out->item_format = (uint32_t*)std::malloc( (totalItems+1)*2*sizeof(uint32_t));
		if (!out->result_format || !out->item_format) throw std::bad_alloc();
	}
	uint64_t rp = 0, ip = 0;
	// `exclusive` option: covered results are dropped on the way out (src/patternMatcher.cpp:192-246, :278-289)
	const bool exclusive = c->inst->compiler.exclusive();
	const uint32_t maxResultSize = c->inst->compiler.maxResultSize();
	std::vector<char> covered;
	for (size_t di=0; di<ndocs; ++di)
	{
		out->doc_result_offsets[ di] = rp;
		const uint64_t b = range[ 2*di], n = range[ 2*di+1];
		if (exclusive)
		{
			covered.assign( n, 0);
			for (uint64_t ai=0; ai<n; ++ai)
			{
				const sp_result_t& r = raw[ b+ai];
				for (uint64_t ni=ai; ni<n; ++ni)
				{
					const sp_result_t& f = raw[ b+ni];
					if (f.origseg > r.origendseg || f.origpos >= r.origend + maxResultSize) break;
					bool differ = (f.origendseg != r.origendseg || f.origend != r.origend || f.origseg != r.origseg || f.origpos != r.origpos);
					if (f.origseg <= r.origseg && f.origpos <= r.origpos && f.origendseg >= r.origendseg && f.origend >= r.origend && differ) covered[ ai] = 1;
					if (f.origseg >= r.origseg && f.origpos >= r.origpos && f.origendseg <= r.origendseg && f.origend <= r.origend && differ) covered[ ni] = 1;
				}
			}
		}
		for (uint64_t ri=0; ri<n; ++ri)
		{
			if (exclusive && covered[ ri]) continue;
			sp_result_t r = raw[ b+ri];
			uint32_t ib = r.item_begin, ic = r.item_count;
			r.item_begin = (uint32_t)ip;
			if (c->withFormats)
			{
				out->result_format[ rp] = rawrf[ b+ri];
				for (uint32_t k=0; k<ic; ++k) { out->item_format[ 2*(ip+k)] = rawif[ 2*(ib+k)]; out->item_format[ 2*(ip+k)+1] = rawif[ 2*(ib+k)+1]; }
			}
			for (uint32_t k=0; k<ic; ++k) out->items[ ip++] = rawitems[ ib+k];
			out->results[ rp++] = r;
		}
	}
	out->doc_result_offsets[ ndocs] = rp;
	out->nresults = rp; out->nitems = ip;
}

} // namespace

extern "C" {

// copies the device results of the last batch to the host, grouped by document
int sp_matcher_ctx_batch_fetch( sp_matcher_ctx_t* c, sp_match_batch_t* out)
{
	std::memset( out, 0, sizeof(*out));
	return guardedCall( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		size_t ndocs = c->lastNdocs;
		uint64_t counters[ SPC_COUNT];
		copySync( c, counters, c->dCounters.ptr, sizeof(counters), hipMemcpyDeviceToHost);
		copyOutBatch( c, 0, ndocs, counters, out);
	});
}

// the same for the documents [first_doc, first_doc+ndocs) of the last batch only
int sp_matcher_ctx_batch_fetch_docs( sp_matcher_ctx_t* c, size_t first_doc, size_t ndocs, sp_match_batch_t* out)
{
	std::memset( out, 0, sizeof(*out));
	return guardedCall( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		if (first_doc > c->lastNdocs || ndocs > c->lastNdocs - first_doc) throw std::runtime_error( "document range outside the last batch");
		uint64_t counters[ SPC_COUNT];
		copySync( c, counters, c->dCounters.ptr, sizeof(counters), hipMemcpyDeviceToHost);
		copyOutBatch( c, first_doc, ndocs, counters, out);
	});
}

int sp_matcher_ctx_batch_status( sp_matcher_ctx_t* c, int32_t* status, size_t ndocs)
{
	return guardedCall( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		if (ndocs > c->lastNdocs) ndocs = c->lastNdocs;
		if (ndocs) copySync( c, status, c->dDocStatus.ptr, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
	});
}

int sp_matcher_ctx_grow_arena( sp_matcher_ctx_t* c)
{
	if (c->arena.maxRules >= (1u<<20)) { c->lasterror = "arena at its maximum size"; return SP_ERR_INVALID; }
	c->arena.maxRules *= 2; c->arena.maxTrigs *= 2; c->arena.bucketCap *= 2; c->arena.maxItems *= 2;
	c->arena.maxRefs *= 2; c->arena.maxFollow *= 2; c->arena.maxDispose *= 2; c->arena.maxHeap *= 2;
	c->arena.maxStaged *= 2; c->arena.maxGStack *= 2; c->arena.winCap *= 2;
	if (c->arena.scratchCap < 256) c->arena.scratchCap = 256;
	c->arenaWaves = 0;
	return SP_OK;
}

int sp_matcher_ctx_reserve_output( sp_matcher_ctx_t* c, uint64_t results, uint64_t items)
{
	if (results > c->minResultCapacity) c->minResultCapacity = results;
	if (items > c->minItemCapacity) c->minItemCapacity = items;
	return SP_OK;
}

} // extern "C"

namespace {

// enqueue one batch on `stream`; all inputs are device pointers
// `rerun` (host entry points): only these documents of the batch already in the output buffers run again, on the
// general kernel in list mode with the working set the caller has just grown -- the results of the other
// documents stay where they are
void launchBatch( sp_matcher_ctx* c, const void* d_lexems, const void* d_origseg, const void* d_doc_offsets,
		  size_t ndocs, size_t nlexems, hipStream_t stream, const void* d_doc_ranges=0, const std::vector<uint32_t>* rerun=0)
{
	HIP_CHECK( hipSetDevice( c->device));
	// geometry: one wave per workgroup; as many as keep every CU busy, never more waves than documents
	size_t waveSlots = (size_t)c->numCUs*SPA_L2_WAVES_PER_CU;
	const size_t ndocsToRun = rerun ? rerun->size() : ndocs;
	unsigned wavesWanted = (unsigned)(ndocsToRun < waveSlots ? ndocsToRun : waveSlots);
	unsigned nblocks = wavesWanted;		// workgroups are single waves
	if (nblocks == 0) nblocks = 1;
	unsigned nwaves = nblocks;
	layoutArena( c->arena);
	{
		// keep the arena below ~48 GiB: fewer resident waves when documents need a large working set
		size_t perWave = (size_t)c->arena.totalWords * sizeof(uint32_t);
		size_t maxWaves = ((size_t)48 << 30) / perWave;
		if (maxWaves < 4) maxWaves = 4;
		if (nwaves > maxWaves) { nblocks = (unsigned)maxWaves; nwaves = nblocks; }
	}
	if (c->arenaWaves < nwaves)
	{
		if (getenv( "SPA_L2_VERBOSE")) fprintf( stderr, "[spa] arena: rules %u bucket %u items %u refs %u staged %u winCap %u -> %.2f MB per wave, %u waves\n", c->arena.maxRules, c->arena.bucketCap, c->arena.maxItems, c->arena.maxRefs, c->arena.maxStaged, c->arena.winCap, c->arena.totalWords*4/1e6, nwaves);
		size_t perWave = (size_t)c->arena.totalWords * sizeof(uint32_t);
		size_t full = (size_t)c->numCUs*SPA_L2_WAVES_PER_CU;
		if (full * perWave > ((size_t)48 << 30)) full = ((size_t)48 << 30) / perWave;
		// batches of many documents: for the full machine at once; a context that sees single documents (the plugin path: one context per
		// host thread) keeps a small arena -- the full one is gigabytes per context
		unsigned alloc = (nwaves >= 64 && nwaves < full && !rerun) ? (unsigned)full : nwaves;
		c->arenaWaves = 0;
		c->dArena.alloc( (size_t)alloc * perWave);
		c->arenaWaves = alloc;
	}
	// output capacity: results are bounded by what fits; sized from the input, grown by the caller on SP_DOC_ERR_ARENA
	uint64_t wantResults = (uint64_t)nlexems*2 + 1024;
	if (wantResults < c->minResultCapacity) wantResults = c->minResultCapacity;
	// item indices in a result record are 32 bit: a batch that needs more fails with SP_DOC_ERR_OUTPUT instead of wrapping
	if (wantResults > 0xFFFFFFFFull) wantResults = 0xFFFFFFFFull;
	if (c->resultCapacity < wantResults)
	{
		c->resultCapacity = 0;		// the capacity follows the buffer: a failed allocation leaves {NULL, 0}, never {NULL, old capacity}
		c->dResults.alloc( wantResults*sizeof(sp_result_t));
		c->resultCapacity = wantResults;
	}
	uint64_t wantItems = (uint64_t)nlexems*6 + 1024;
	if (wantItems < c->minItemCapacity) wantItems = c->minItemCapacity;
	if (wantItems > 0xFFFFFFFFull) wantItems = 0xFFFFFFFFull;
	if (c->itemCapacity < wantItems)
	{
		c->itemCapacity = 0;
		c->dItems.alloc( wantItems*sizeof(sp_result_item_t));
		c->itemCapacity = wantItems;
	}
	if (c->withFormats)
	{
		c->dResultFormat.reserve( c->resultCapacity*sizeof(uint32_t));
		c->dItemFormat.reserve( c->itemCapacity*2*sizeof(uint32_t));
	}
	c->dDocRange.reserve( (ndocs+1)*2*sizeof(uint64_t));
	c->dDocStats.reserve( (ndocs+1)*4*sizeof(uint64_t));
	c->dDocStatus.reserve( (ndocs+1)*sizeof(int32_t));

	HIP_CHECK( hipMemsetAsync( c->dCursor.ptr, 0, 256, stream));
	if (!rerun) HIP_CHECK( hipMemsetAsync( c->dCounters.ptr, 0, SPC_COUNT*sizeof(uint64_t), stream));
	else HIP_CHECK( hipMemsetAsync( (uint64_t*)c->dCounters.ptr + SPC_FAILED, 0, sizeof(uint64_t), stream));	// (the other counters continue)

	L2Params P;
	std::memset( &P, 0, sizeof(P));
	P.programs = (const DevProgram*)c->dPrograms.ptr;
	P.trigdefs = (const DevTrigDef*)c->dTrigdefs.ptr;
	P.keytab = (const DevKeyEntry*)c->dKeytab.ptr;
	P.keylist = (const DevKeyRef*)c->dKeylist.ptr;
	P.keymask = c->keymask; P.nofStopWords = c->nofStopWords;
	P.lexems = (const uint32_t*)d_lexems; P.origseg = (const uint32_t*)d_origseg;
	P.docOffsets = (const uint64_t*)d_doc_offsets;
	P.docRangesIn = (const uint64_t*)d_doc_ranges;
	P.ndocs = (uint32_t)ndocs; P.withItems = c->withItems ? 1u : 0u;
	P.arenaBase = (uint32_t*)c->dArena.ptr; P.arena = c->arena;
	P.docCursor = (uint32_t*)c->dCursor.ptr;
	P.counters = (uint64_t*)c->dCounters.ptr;
	P.results = (uint32_t*)c->dResults.ptr; P.resultCapacity = c->resultCapacity;
	P.items = (uint32_t*)c->dItems.ptr; P.itemCapacity = c->itemCapacity;
	P.docRange = (uint64_t*)c->dDocRange.ptr;
	P.docStats = (uint64_t*)c->dDocStats.ptr;
	P.docStatus = (int32_t*)c->dDocStatus.ptr;
	P.withFormats = c->withFormats ? 1u : 0u;
	P.resultFormat = (uint32_t*)c->dResultFormat.ptr; P.itemFormat = (uint32_t*)c->dItemFormat.ptr;

#if defined(SPA_TRACE) || defined(SPA_POLL)
	static uint32_t* traceHost = 0;
	if (!traceHost && (getenv("SPA_HOSTALLOC") || 
#ifdef SPA_TRACE
		1
#else
		0
#endif
		))
	{
		HIP_CHECK( hipHostMalloc( (void**)&traceHost, 4096, hipHostMallocMapped));
		std::memset( traceHost, 0xEE, 4096);
	}
	void* traceDev = 0;
	if (traceHost) HIP_CHECK( hipHostGetDevicePointer( &traceDev, traceHost, 0));
	P.trace = (uint32_t*)traceDev;
#endif
	HIP_CHECK( hipEventRecord( c->evStart, stream));
	if (rerun)
	{
		const uint32_t n = (uint32_t)rerun->size();
		c->dFallbackList.reserve( (ndocs+1)*sizeof(uint32_t));
		HIP_CHECK( hipMemcpyAsync( c->dFallbackList.ptr, rerun->data(), n*sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		HIP_CHECK( hipMemcpyAsync( (uint32_t*)c->dCursor.ptr + 2, &n, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		HIP_CHECK( hipStreamSynchronize( stream));		// (the list and its count are host temporaries)
		P.docList = (const uint32_t*)c->dFallbackList.ptr; P.docListCount = (const uint32_t*)c->dCursor.ptr + 2;
		P.docCursor = (uint32_t*)c->dCursor.ptr + 1;
		HIP_CHECK( launchL2Match( P, nblocks, stream));
	}
	else if (c->join)
	{
		// opt-in prototype: result sets by joining positions, nothing installed (l2_join.h)
		JoinParams J;
		std::memset( &J, 0, sizeof(J));
		c->dJoinCounts.reserve( (nlexems + 64) * sizeof(uint32_t));
		J.filter = (const uint32_t*)c->dJoinFilter.ptr; J.counts = (uint32_t*)c->dJoinCounts.ptr; J.countsCapacity = nlexems;
		J.keytab = (const JoinKey*)c->dJoinKeytab.ptr; J.keymask = c->joinKeymask; J.rules = (const JoinRule*)c->dJoinRules.ptr; J.maxRange = c->joinMaxRange; J.delimiter = c->joinDelimiter;
		J.lexems = P.lexems; J.origseg = P.origseg; J.docOffsets = P.docOffsets; J.docRangesIn = P.docRangesIn; J.ndocs = P.ndocs;
		J.docCursor = (uint32_t*)c->dCursor.ptr;
		J.counters = P.counters; J.results = P.results; J.resultCapacity = P.resultCapacity;
		J.items = P.items; J.itemCapacity = P.itemCapacity; J.withItems = P.withItems; J.itemFormat = P.itemFormat;
		J.docRange = P.docRange; J.docStats = P.docStats; J.docStatus = P.docStatus;
		J.withFormats = P.withFormats; J.resultFormat = P.resultFormat;
		const size_t jslots = (size_t)c->numCUs * 32;		// one wave per document, no LDS, few registers
		HIP_CHECK( launchL2Join( J, (unsigned)(ndocs < jslots ? (ndocs ? ndocs : 1) : jslots), stream));
	}
	else if (c->fast)
	{
		// flat rule set: the LDS-resident kernel first; the documents it hands over (fallbackList) go through the
		// general kernel in list mode right behind it on the same stream (an empty list costs one short launch)
		uint32_t fR = 0, fT = 0;
		fastCapacities( c->fastVariant, fR, fT);
		layoutFast( c->fastSpill, c->fastBucketMeta, c->fastExpShift, c->fastKeyinst, fR, fT, c->fastMaxRules, c->fastMaxStaged);
		if (!c->fastBlocksPerCU) c->fastBlocksPerCU = (unsigned)fastBlocksPerCU( c->fastVariant);
		size_t fslots = (size_t)c->numCUs * c->fastBlocksPerCU;
		unsigned fblocks = (unsigned)(ndocs < fslots ? ndocs : fslots);
		if (fblocks == 0) fblocks = 1;
		if (c->fastWaves < fblocks)
		{
			c->fastWaves = 0;
			const size_t spillSlots = fblocks >= 64 ? fslots : fblocks;		// (single documents: a small spill area, see the arena)
			c->dSpill.alloc( spillSlots * (size_t)c->fastSpill.totalWords * sizeof(uint32_t));
			c->fastWaves = (unsigned)spillSlots;
			if (getenv( "SPA_L2_VERBOSE")) fprintf( stderr, "[spa] fast tier: %u waves/CU, LDS capacities R %u T %u, spill %.2f MB per wave\n", c->fastBlocksPerCU, fR, fT, c->fastSpill.totalWords*4/1e6);
		}
		c->dFallbackList.reserve( (ndocs+1)*sizeof(uint32_t));
		FastParams F;
		std::memset( &F, 0, sizeof(F));
		F.keyinst = (const FastKeyInst*)c->dKeyinst.ptr; F.statics = (const FastStatic*)c->dStatics.ptr; F.keytab = (const FastKeyEntry*)c->dKeytab.ptr;
		F.keymask = c->keymask; F.nofStopWords = c->nofStopWords;
		F.lexems = P.lexems; F.origseg = P.origseg; F.docOffsets = P.docOffsets; F.docRangesIn = P.docRangesIn;
		F.ndocs = P.ndocs; F.withItems = P.withItems;
		std::memcpy( F.bucketMeta, c->fastBucketMeta, sizeof(F.bucketMeta)); F.expShift = c->fastExpShift;
		F.spill = c->fastSpill; F.spillBase = (uint32_t*)c->dSpill.ptr;
		F.docCursor = (uint32_t*)c->dCursor.ptr;
		F.counters = P.counters; F.results = P.results; F.resultCapacity = P.resultCapacity; F.items = P.items; F.itemCapacity = P.itemCapacity;
		F.docRange = P.docRange; F.docStats = P.docStats; F.docStatus = P.docStatus;
		F.withFormats = P.withFormats; F.resultFormat = P.resultFormat; F.itemFormat = P.itemFormat;
		F.fallbackList = (uint32_t*)c->dFallbackList.ptr; F.fallbackCount = (uint32_t*)c->dCursor.ptr + 2;
		F.diag = (uint32_t*)c->dCursor.ptr + 16; F.prof = (uint64_t*)((uint32_t*)c->dCursor.ptr + 32);
		HIP_CHECK( launchL2Fast( F, c->fastVariant, fblocks, stream));
		P.docList = F.fallbackList; P.docListCount = F.fallbackCount;
		P.docCursor = (uint32_t*)c->dCursor.ptr + 1;
		const unsigned listBlocks = nblocks < 2*c->numCUs ? nblocks : 2*c->numCUs;
		HIP_CHECK( launchL2Match( P, listBlocks, stream));
	}
	else
	{
		HIP_CHECK( launchL2Match( P, nblocks, stream));
	}
	HIP_CHECK( hipEventRecord( c->evStop, stream));
#if defined(SPA_TRACE) || defined(SPA_POLL)
	static uint32_t traceDummy[16];
	uint32_t* traceShow = traceHost ? traceHost : traceDummy;
	for (int waited=0; hipStreamQuery( stream) == hipErrorNotReady; ++waited)
	{
		usleep( 100000);
		if (waited == 100 || waited == 150)
		{
			fprintf( stderr, "[spa trace] kernel still running after %d ms:", waited*100);
			for (int i=0; i<16; ++i) fprintf( stderr, " [%d]=%u", i, traceShow[i]);
			fprintf( stderr, "\n");
			if (waited == 150) { fflush( stderr); _exit( 3); }
		}
	}
#endif
	c->evValid = true; c->lastStream = stream; c->lastNdocs = ndocs;
}

} // namespace

extern "C" {

int sp_matcher_ctx_match_docs_device( sp_matcher_ctx_t* c, const void* d_lexems, const void* d_origseg,
				      const void* d_doc_offsets, size_t ndocs, size_t nlexems,
				      void* stream, sp_match_device_batch_t* out)
{
	return guardedCall( c->lasterror, SP_ERR_INVALID, [&]{
		if (ndocs >= 0xFFFFFFFFull) throw std::runtime_error( "too many documents in one batch");
		launchBatch( c, d_lexems, d_origseg, d_doc_offsets, ndocs, nlexems, (hipStream_t)stream);
		if (out)
		{
			out->ndocs = ndocs;
			out->d_results = c->dResults.ptr; out->d_items = c->dItems.ptr;
			out->d_doc_result_offsets = c->dDocRange.ptr;
			out->d_doc_stats = c->dDocStats.ptr; out->d_doc_status = c->dDocStatus.ptr;
			out->d_counters = c->dCounters.ptr;
			out->d_result_format = c->withFormats ? c->dResultFormat.ptr : 0;
			out->d_item_format = c->withFormats ? c->dItemFormat.ptr : 0;
		}
	});
}

int sp_matcher_ctx_match_lexed_device( sp_matcher_ctx_t* c, const void* d_lexems, const void* d_doc_ranges,
				       size_t ndocs, size_t nlexems_hint, void* stream, sp_match_device_batch_t* out)
{
	return guardedCall( c->lasterror, SP_ERR_INVALID, [&]{
		if (ndocs >= 0xFFFFFFFFull) throw std::runtime_error( "too many documents in one batch");
		launchBatch( c, d_lexems, 0, 0, ndocs, nlexems_hint, (hipStream_t)stream, d_doc_ranges);
		if (out)
		{
			out->ndocs = ndocs;
			out->d_results = c->dResults.ptr; out->d_items = c->dItems.ptr;
			out->d_doc_result_offsets = c->dDocRange.ptr;
			out->d_doc_stats = c->dDocStats.ptr; out->d_doc_status = c->dDocStatus.ptr;
			out->d_counters = c->dCounters.ptr;
			out->d_result_format = c->withFormats ? c->dResultFormat.ptr : 0;
			out->d_item_format = c->withFormats ? c->dItemFormat.ptr : 0;
		}
	});
}

int sp_matcher_ctx_batch_counters( sp_matcher_ctx_t* c, uint64_t counters[8])
{
	return guardedCall( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		copySync( c, counters, c->dCounters.ptr, SPC_COUNT*sizeof(uint64_t), hipMemcpyDeviceToHost);
		if (c->fast && getenv( "SPA_L2_VERBOSE"))
		{
			uint32_t diag[ 16];
			copySync( c, diag, (const uint32_t*)c->dCursor.ptr + 16, sizeof(diag), hipMemcpyDeviceToHost);
#ifdef SPA_PROF
			uint64_t prof[ 12];
			copySync( c, prof, (const uint32_t*)c->dCursor.ptr + 32, sizeof(prof), hipMemcpyDeviceToHost);
			double tot = 0; for (int i=0; i<6; ++i) tot += (double)prof[ i];
			
			fprintf( stderr, "[spa] fast tier phases (share of wave cycles): scan+fire %.1f%% install %.1f%% deactivate %.1f%% expiry %.1f%% results %.1f%% fetch %.1f%% | inside deactivation: loads %.1f%% ranks+queue %.1f%% replay %.1f%% (%.2f replay steps, %.2f batches, %.2f rules per event); %.0f cycles per event\n",
				100*prof[0]/tot, 100*prof[1]/tot, 100*prof[2]/tot, 100*prof[3]/tot, 100*prof[4]/tot, 100*prof[5]/tot, 100*prof[6]/tot, 100*prof[7]/tot, 100*prof[11]/tot,
				(double)prof[ 8] / (double)(counters[ SPC_EVENTS] ? counters[ SPC_EVENTS] : 1), (double)prof[ 9] / (double)(counters[ SPC_EVENTS] ? counters[ SPC_EVENTS] : 1), (double)prof[ 10] / (double)(counters[ SPC_EVENTS] ? counters[ SPC_EVENTS] : 1),
				tot / (double)(counters[ SPC_EVENTS] ? counters[ SPC_EVENTS] : 1));
#endif
			if (diag[ 0])
			{
				fprintf( stderr, "[spa] fast tier handed %u of %zu documents to the general kernel; by reason:", diag[ 0], c->lastNdocs);
				for (int i=1; i<16; ++i) if (diag[ i]) fprintf( stderr, " [%d]=%u", i, diag[ i]);
				fprintf( stderr, "\n");
			}
		}
	});
}

double sp_matcher_ctx_last_kernel_ms( sp_matcher_ctx_t* c)
{
	if (!c->evValid) return -1.0;
	float ms = 0.0f;
	if (hipEventSynchronize( c->evStop) != hipSuccess) return -1.0;
	if (hipEventElapsedTime( &ms, c->evStart, c->evStop) != hipSuccess) return -1.0;
	return (double)ms;
}

int sp_matcher_ctx_match_docs( sp_matcher_ctx_t* c, const sp_lexem_t* lexems, const uint32_t* origseg,
			       const uint64_t* doc_offsets, size_t ndocs, sp_match_batch_t* out)
{
	std::memset( out, 0, sizeof(*out));
	return guardedCall( c->lasterror, SP_ERR_INVALID, [&]{
		if (ndocs >= 0xFFFFFFFFull) throw std::runtime_error( "too many documents in one batch");
		HIP_CHECK( hipSetDevice( c->device));
		size_t nlex = ndocs ? (size_t)doc_offsets[ ndocs] : 0;
		c->dLexems.reserve( (nlex+1)*sizeof(sp_lexem_t));
		c->dDocOffsets.reserve( (ndocs+1)*sizeof(uint64_t));
		if (nlex) copySync( c, c->dLexems.ptr, lexems, nlex*sizeof(sp_lexem_t), hipMemcpyHostToDevice);
		copySync( c, c->dDocOffsets.ptr, doc_offsets, (ndocs+1)*sizeof(uint64_t), hipMemcpyHostToDevice);
		const void* dseg = 0;
		if (origseg)
		{
			c->dOrigseg.reserve( (nlex+1)*sizeof(uint32_t));
			if (nlex) copySync( c, c->dOrigseg.ptr, origseg, nlex*sizeof(uint32_t), hipMemcpyHostToDevice);
			dseg = c->dOrigseg.ptr;
		}
		uint64_t counters[ SPC_COUNT];
		std::vector<uint32_t> again;		// documents whose working set exceeded the per-wave arena: only they run again
		for (int attempt=0;; ++attempt)
		{
			launchBatch( c, c->dLexems.ptr, dseg, c->dDocOffsets.ptr, ndocs, nlex, c->own, 0, again.empty() ? 0 : &again);
			HIP_CHECK( hipStreamSynchronize( c->own));
			copySync( c, counters, c->dCounters.ptr, sizeof(counters), hipMemcpyDeviceToHost);
			// the output counters keep counting past the capacity: if the buffers were too small,
			// grow them to what this batch needs and run it again (the kernel is deterministic)
			bool grow = false;
			again.clear();
			if (counters[ SPC_RESULTS] > c->resultCapacity) { c->minResultCapacity = counters[ SPC_RESULTS] + counters[ SPC_RESULTS]/8 + 1024; grow = true; }
			if (counters[ SPC_ITEMS] > c->itemCapacity) { c->minItemCapacity = counters[ SPC_ITEMS] + counters[ SPC_ITEMS]/8 + 1024; grow = true; }
			if (counters[ SPC_FAILED])
			{
				// documents whose working set exceeded the per-wave arena: double the arena and run THEM again
				// (the whole batch only when the output buffers have to be reallocated as well)
				std::vector<int32_t> st( ndocs);
				copySync( c, st.data(), c->dDocStatus.ptr, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
				std::vector<uint32_t> arenaDocs;
				for (size_t di=0; di<ndocs; ++di) if (st[ di] == SPD_ERR_ARENA) arenaDocs.push_back( (uint32_t)di);
				if (!arenaDocs.empty() && sp_matcher_ctx_grow_arena( c) == SP_OK)
				{
					if (!grow) again.swap( arenaDocs);
					grow = true;
				}
			}
			if (!grow || attempt >= 12) break;
		}
		{
			// a partial rerun counts failures among the documents it ran again only: the batch's count is what the statuses say
			std::vector<int32_t> st( ndocs+1);
			if (ndocs) copySync( c, st.data(), c->dDocStatus.ptr, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
			uint64_t failed = 0;
			for (size_t di=0; di<ndocs; ++di) if (st[ di] != 0) ++failed;
			if (failed != counters[ SPC_FAILED])
			{
				counters[ SPC_FAILED] = failed;
				copySync( c, (uint64_t*)c->dCounters.ptr + SPC_FAILED, &failed, sizeof(failed), hipMemcpyHostToDevice);
			}
		}
		copyOutBatch( c, 0, ndocs, counters, out);
		if (counters[ SPC_FAILED])
		{
			size_t bad = 0;
			while (bad < ndocs && out->doc_status[ bad] == 0) ++bad;
			char msg[ 128];
			snprintf( msg, sizeof(msg), "at least one document failed: document %zu has status %d (see doc_status)", bad, bad < ndocs ? out->doc_status[ bad] : -1);
			throw std::runtime_error( msg);
		}
	}) == SP_OK ? SP_OK : (c->lasterror.find( "document failed") != std::string::npos ? SP_ERR_MATCH : SP_ERR_DEVICE);
}

void sp_match_batch_free( sp_match_batch_t* b)
{
	std::free( b->results); std::free( b->items); std::free( b->doc_result_offsets);
	std::free( b->doc_stats); std::free( b->doc_status); std::free( b->result_format); std::free( b->item_format);
	std::memset( b, 0, sizeof(*b));
}

// ---- single-document mode: PatternMatcherContextInterface ----
int sp_matcher_ctx_put_input( sp_matcher_ctx_t* c, const sp_lexem_t* lexems, const uint32_t* origseg, size_t n)
{
	return guardedCall( c->lasterror, SP_ERR_INVALID, [&]{
		// ascending-order contract checked up front like the reference does per call (src/patternMatcher.cpp:136-139)
		uint32_t cur = c->curLexems.empty() ? 0 : c->curLexems.back().ordpos;
		for (size_t i=0; i<n; ++i)
		{
			if (lexems[i].ordpos < cur) throw std::runtime_error( "term events not fed in ascending order");
			cur = lexems[i].ordpos;
		}
		if (origseg && !c->curHasSeg) { c->curOrigseg.assign( c->curLexems.size(), 0); c->curHasSeg = true; }
		c->curLexems.insert( c->curLexems.end(), lexems, lexems+n);
		if (c->curHasSeg)
		{
			if (origseg) c->curOrigseg.insert( c->curOrigseg.end(), origseg, origseg+n);
			else c->curOrigseg.insert( c->curOrigseg.end(), n, 0u);
		}
	});
}

int sp_matcher_ctx_fetch_results( sp_matcher_ctx_t* c, sp_result_t** results, size_t* nresults,
				  sp_result_item_t** items, size_t* nitems)
{
	uint64_t offs[2] = {0, (uint64_t)c->curLexems.size()};
	sp_match_batch_t b;
	sp_lexem_t dummy = {0,0,0,0};
	int rc = sp_matcher_ctx_match_docs( c, c->curLexems.empty() ? &dummy : c->curLexems.data(),
					c->curHasSeg ? c->curOrigseg.data() : 0, offs, 1, &b);
	if (rc != SP_OK && rc != SP_ERR_MATCH) { sp_match_batch_free( &b); return rc; }
	if (rc == SP_ERR_MATCH)
	{
		static const char* msg[] = {"ok", "term events not fed in ascending order", "working set of the document exceeds the arena",
			"pattern with too many identical key events defined", "internal: encountered past trigger with follow",
			"term event out of range", "illegal free of event data reference"};
		int st = b.doc_status ? b.doc_status[0] : 0;
		c->lasterror = std::string("failed to feed input to pattern matcher: ") + ((st >= 0 && st <= 6) ? msg[ st] : "unknown");
		sp_match_batch_free( &b);
		return SP_ERR_MATCH;
	}
	c->lastStats.nofProgramsInstalled = (double)b.doc_stats[0];
	c->lastStats.nofAltKeyProgramsInstalled = (double)b.doc_stats[1];
	c->lastStats.nofSignalsFired = (double)b.doc_stats[2];
	c->lastStats.nofTriggersAvgActive = c->curLexems.empty() ? 0.0 : (double)b.doc_stats[3] / (double)c->curLexems.size();
	c->curResultFormat.clear(); c->curItemFormat.clear();
	if (b.result_format) c->curResultFormat.assign( b.result_format, b.result_format + b.nresults);
	if (b.item_format) c->curItemFormat.assign( b.item_format, b.item_format + 2*b.nitems);
	*results = b.results; *nresults = b.nresults; b.results = 0;
	if (items) { *items = b.items; b.items = 0; }
	if (nitems) *nitems = b.nitems;
	sp_match_batch_free( &b);
	return SP_OK;
}

int sp_matcher_ctx_fetch_formats( sp_matcher_ctx_t* c, const uint32_t** result_format, const uint32_t** item_format)
{
	*result_format = c->withFormats ? c->curResultFormat.data() : 0;
	*item_format = c->withFormats ? c->curItemFormat.data() : 0;
	return SP_OK;
}

int sp_matcher_ctx_statistics( sp_matcher_ctx_t* c, sp_matcher_stats_t* out)
{
	*out = c->lastStats;
	return SP_OK;
}

int sp_matcher_ctx_reset( sp_matcher_ctx_t* c)
{
	c->curLexems.clear(); c->curOrigseg.clear(); c->curHasSeg = false;
	c->curResultFormat.clear(); c->curItemFormat.clear();
	std::memset( &c->lastStats, 0, sizeof(c->lastStats));
	return SP_OK;
}

} // extern "C"
