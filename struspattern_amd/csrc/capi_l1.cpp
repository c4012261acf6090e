// C-ABI of level 1 (include/strus_pattern_amd.h): lexer compiler handle + GPU lexer context.
// No CPU fallback: a context cannot be created without a usable HIP device.
#include "../../include/strus_pattern_amd.h"
#include "l1_compile.hpp"
#include "l1_device.h"
#include "hip_util.hpp"
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace spa {
bool l1ScanByLanes( const L1Params& PS, const L1Params& P);
hipError_t launchL1Lex( const L1Params& PS, const L1Params& PW, const L1Params& P, unsigned nblocks, unsigned nthreads, unsigned laneBlocks, unsigned wordBlocks, unsigned wordWaves, unsigned postWaves, hipStream_t stream, hipEvent_t betweenKernels, hipEvent_t afterWords);
}
using namespace spa;

struct sp_lexer
{
	LexCompiler compiler;
	mutable std::string lasterror;
};

namespace {
template <class FN>
int guardedCall1( std::string& err, int errcode, FN fn)
{
	try { fn(); return SP_OK; }
	catch (const std::bad_alloc&) { err = "memory allocation error in strus pattern"; return SP_ERR_NOMEM; }
	catch (const HipError& e) { err = e.what(); return SP_ERR_DEVICE; }
	catch (const std::exception& e) { err = e.what(); return errcode; }
}
}

struct sp_lexer_ctx
{
	const sp_lexer* inst;
	int device;
	std::string lasterror;
	DeviceBuffer dByteClass, dClassCtx, dCharMask, dStartMask, dAcceptMask, dShiftDst, dSelfLoop, dExSrc, dExDst, dExCount,
		dPatterns, dSymbols, dSymbolText, dLiterals, dLiteralText, dLitPats, dTableImage, dWordsImage, dPatOfBit, dApprox, dCharCp, dCharPos, dCpBlocks, dCpPages, dUnitStart, dDocSequential, dNullable,
		dScanImage, dShapes, dShapePats;
	uint32_t ldsWords, ldsAccept, ldsStart, ldsShift, ldsSelf, ldsExSrc, ldsExDst; unsigned blockThreads;
	uint32_t wChar, wAccept, wStart, wShift, wSelf, wExSrc, wExDst, wShapeFp, wWords;	// the words kernel's image: the passes behind the scanned ones + the shape table; offsets biased by what is left out
	uint32_t imgShapeFp, imgWords, imgAccept, imgStart, imgShift, imgSelf, imgExSrc, imgExDst;	// offsets inside the image of all passes (dTableImage); lds*: inside the image of the scanned passes
	bool wordsKernel;		// plain tables: literals and word shapes are found by the words kernel
	DeviceBuffer dArena, dCounters, dText, dDocOffsets, dLexems, dDocRange, dDocStatus, dQueue, dReportCount, dWordQueue, dWordCount;
	uint32_t queueMul;		// report queue between the two kernels: queueMul/16 reports per text byte (+64 per document)
	uint32_t queueCap, eventCap;
	unsigned arenaWaves; uint64_t arenaWords;
	uint64_t lexemCapacity, minLexemCapacity;
	unsigned numCUs;
	hipEvent_t evStart, evMid, evWords, evStop; bool evValid;
	char scanKernel[ 48] = "(none)";
	const char* wordsKernelName = "(none)";
	hipStream_t lastStream; size_t lastNdocs;
	hipStream_t own;		// the context's own stream (non-blocking): the host-buffer entry points of different contexts -- one per host thread,
				// the reference's threading model -- copy and launch side by side instead of queueing on the null stream
	sp_lexer_ctx() :inst(0),device(0),ldsWords(0),ldsAccept(0),ldsStart(0),ldsShift(0),ldsSelf(0),ldsExSrc(0),ldsExDst(0),blockThreads(256),queueMul(8),queueCap(4096),eventCap(32768),arenaWaves(0),arenaWords(0),lexemCapacity(0),minLexemCapacity(0)
		,numCUs(256),evStart(0),evMid(0),evWords(0),evStop(0),evValid(false),lastStream(0),lastNdocs(0),own(0)
		,wChar(0),wAccept(0),wStart(0),wShift(0),wSelf(0),wExSrc(0),wExDst(0),wShapeFp(0),wWords(0),imgShapeFp(0),imgWords(0),imgAccept(0),imgStart(0),imgShift(0),imgSelf(0),imgExSrc(0),imgExDst(0),wordsKernel(false){}
};

extern "C" {

sp_lexer_t* sp_lexer_create(void) { try { return new sp_lexer(); } catch (...) { return 0; } }
void sp_lexer_free( sp_lexer_t* l) { delete l; }
const char* sp_lexer_last_error( const sp_lexer_t* l) { return l->lasterror.c_str(); }

#define LGUARD( CODE, BODY) return guardedCall1( l->lasterror, CODE, [&]{ BODY; })

// the compiled lexer as a blob (automaton tables, literal and symbol tables, names) and back: SURVEY.md 8(f).4
int sp_lexer_serialize( const sp_lexer_t* l, void** blob, size_t* size)
{
	*blob = 0; *size = 0;
	return guardedCall1( l->lasterror, SP_ERR_INVALID, [&]{
		std::vector<uint8_t> buf;
		l->compiler.save( buf);
		*blob = std::malloc( buf.size() ? buf.size() : 1);
		if (!*blob) throw std::bad_alloc();
		std::memcpy( *blob, buf.data(), buf.size());
		*size = buf.size();
	});
}
sp_lexer_t* sp_lexer_deserialize( const void* blob, size_t size, char* err, size_t errsize)
{
	sp_lexer* l = 0;
	try
	{
		l = new sp_lexer();
		l->compiler.load( blob, size);
		return l;
	}
	catch (const std::exception& e)
	{
		if (err && errsize) { std::strncpy( err, e.what(), errsize-1); err[ errsize-1] = 0; }
		delete l;
		return 0;
	}
}


int sp_lexer_define_lexem_name( sp_lexer_t* l, uint32_t id, const char* name)
{ LGUARD( SP_ERR_INVALID, l->compiler.defineLexemName( id, name ? name : "")); }
const char* sp_lexer_get_lexem_name( const sp_lexer_t* l, uint32_t id) { return l->compiler.getLexemName( id); }
int sp_lexer_define_lexem( sp_lexer_t* l, uint32_t id, const char* expression, uint32_t resultIndex, uint32_t level, int posbind)
{ LGUARD( SP_ERR_INVALID, l->compiler.defineLexem( id, expression ? expression : "", resultIndex, level, posbind)); }
int sp_lexer_define_symbol( sp_lexer_t* l, uint32_t symbolid, uint32_t patternid, const char* name)
{ LGUARD( SP_ERR_INVALID, l->compiler.defineSymbol( symbolid, patternid, name ? name : "")); }
uint32_t sp_lexer_get_symbol( const sp_lexer_t* l, uint32_t patternid, const char* name) { return l->compiler.getSymbol( patternid, name ? name : ""); }
int sp_lexer_define_option( sp_lexer_t* l, const char* name, double value)
{ LGUARD( SP_ERR_INVALID, l->compiler.defineOption( name ? name : "", value)); }
int sp_lexer_compile( sp_lexer_t* l)
{ LGUARD( SP_ERR_COMPILE, l->compiler.compile()); }

// Flat dump for tests: header of 8 words {nofPasses, nofClasses, maxExceptions, nofPatterns, nofPositions, nofLiterals, reportsOrdered, 0},
// byteClass[256], classCtx[nofClasses], charMask, startMask, acceptMask, shiftDst, selfLoop,
// exCount[nofPasses], exSrc, exDst, then per pattern {id, word, levelBind, prefixLen, suffixLen, mask}
size_t sp_lexer_dump_tables( const sp_lexer_t* l, uint64_t** out)
{
	const LexTables& T = l->compiler.tables();
	std::vector<uint64_t> b;
	b.push_back( T.nofPasses); b.push_back( T.nofClasses); b.push_back( T.maxExceptions); b.push_back( T.patterns.size());
	b.push_back( T.nofPositions); b.push_back( 0); b.push_back( 0); b.push_back( T.ucp ? 1 : 0);
	for (size_t i=0; i<T.byteClass.size(); ++i) b.push_back( T.byteClass[i]);
	for (size_t i=0; i<T.classCtx.size(); ++i) b.push_back( T.classCtx[i]);
	b.insert( b.end(), T.charMask.begin(), T.charMask.end());
	b.insert( b.end(), T.startMask.begin(), T.startMask.end());
	b.insert( b.end(), T.acceptMask.begin(), T.acceptMask.end());
	b.insert( b.end(), T.shiftDst.begin(), T.shiftDst.end());
	b.insert( b.end(), T.selfLoop.begin(), T.selfLoop.end());
	for (size_t i=0; i<T.exCount.size(); ++i) b.push_back( T.exCount[i]);
	b.insert( b.end(), T.exSrc.begin(), T.exSrc.end());
	b.insert( b.end(), T.exDst.begin(), T.exDst.end());
	for (size_t i=0; i<T.patterns.size(); ++i)
	{
		const DevLexPattern& p = T.patterns[i];
		b.push_back( p.id); b.push_back( p.word); b.push_back( p.levelBind); b.push_back( p.prefixLen); b.push_back( p.suffixLen);
		b.push_back( ((uint64_t)p.maskHi << 32) | p.maskLo); b.push_back( p.defIndex);
	}
	// whole-word literals: count, then per literal {len, patCount, bytes..., pattern indices...}
	b[5] = T.nofLiterals;
	b[6] = T.reportsOrdered ? 1 : 0;
	for (size_t i=0; i<T.literals.size(); ++i)
	{
		const DevLiteral& e = T.literals[i];
		if (!e.hash) continue;
		b.push_back( e.len); b.push_back( e.patCount);
		for (uint32_t k=0; k<e.len; ++k) b.push_back( T.literalText[ e.textOffset+k]);
		for (uint32_t k=0; k<e.patCount; ++k) b.push_back( T.litPats[ e.patBegin+k]);
	}
	// ALLOWEMPTY: {patterns entry, emptyOk bits} per expression that matches the empty string
	b.push_back( T.nullable.size());
	for (size_t i=0; i<T.nullable.size(); ++i) { b.push_back( T.nullable[ i].pattern); b.push_back( T.nullable[ i].emptyOk); }
	// classes by code point: block table and pages (both may be empty)
	b.push_back( T.cpBlocks.size());
	for (size_t i=0; i<T.cpBlocks.size(); ++i) b.push_back( T.cpBlocks[ i]);
	b.push_back( T.cpPages.size());
	for (size_t i=0; i<T.cpPages.size(); ++i) b.push_back( T.cpPages[ i]);
	// word shapes: passes the scan kernel runs, expressions taken as shapes, then per table entry {tag, key, patCount, pattern indices...}
	b.push_back( T.scanPasses); b.push_back( T.nofShapes);
	for (size_t i=0; i<T.shapes.size(); ++i)
	{
		const DevShape& e = T.shapes[ i];
		if (!e.tag) continue;
		b.push_back( e.tag); b.push_back( e.key); b.push_back( e.patCount);
		for (uint32_t k=0; k<e.patCount; ++k) b.push_back( T.shapePats[ e.patBegin+k]);
	}
	*out = (uint64_t*)std::malloc( (b.size()+1)*sizeof(uint64_t));
	if (!*out) return 0;
	std::memcpy( *out, b.data(), b.size()*sizeof(uint64_t));
	return b.size();
}

sp_lexer_ctx_t* sp_lexer_ctx_create( const sp_lexer_t* l, int device)
{
	sp_lexer_ctx* c = 0;
	try
	{
		if (!l->compiler.compiled())
		{
			l->lasterror = "called create context without calling 'compile'";	// src/patternLexer.cpp:1124-1127
			return 0;
		}
		int ndev = 0;
		hipError_t e = hipGetDeviceCount( &ndev);
		if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
		{
			l->lasterror = "no usable HIP device: the pattern lexer runs on the GPU only (no CPU fallback)";
			return 0;
		}
		c = new sp_lexer_ctx();
		c->inst = l; c->device = device;
		HIP_CHECK( hipSetDevice( device));
		hipDeviceProp_t prop;
		HIP_CHECK( hipGetDeviceProperties( &prop, device));
		c->numCUs = prop.multiProcessorCount > 0 ? (unsigned)prop.multiProcessorCount : 256u;
		const LexTables& T = l->compiler.tables();
		if (T.nofPasses > 32) throw std::runtime_error( "too many regular expression positions for this version (more than 32 passes of 4096 positions)");
		c->dByteClass.upload( T.byteClass.data(), T.byteClass.size());
		c->dClassCtx.upload( T.classCtx.data(), T.classCtx.size());
		c->dCharMask.upload( T.charMask.data(), T.charMask.size()*8);
		c->dStartMask.upload( T.startMask.data(), T.startMask.size()*8);
		c->dAcceptMask.upload( T.acceptMask.data(), T.acceptMask.size()*8);
		c->dShiftDst.upload( T.shiftDst.data(), T.shiftDst.size()*8);
		c->dSelfLoop.upload( T.selfLoop.data(), T.selfLoop.size()*8);
		c->dExSrc.upload( T.exSrc.data(), T.exSrc.size()*8);
		c->dExDst.upload( T.exDst.data(), T.exDst.size()*8);
		c->dExCount.upload( T.exCount.data(), T.exCount.size()*4);
		c->dPatOfBit.upload( T.patOfBit.data(), T.patOfBit.size()*4);
		c->dPatterns.upload( T.patterns.data(), T.patterns.size()*sizeof(DevLexPattern));
		c->dSymbols.upload( T.symbols.data(), T.symbols.size()*sizeof(DevSymbol));
		c->dSymbolText.upload( T.symbolText.data(), T.symbolText.size());
		c->dLiterals.upload( T.literals.data(), T.literals.size()*sizeof(DevLiteral));
		c->dLiteralText.upload( T.literalText.data(), T.literalText.size());
		c->dLitPats.upload( T.litPats.data(), T.litPats.size()*4);
		if (!T.cpBlocks.empty()) { c->dCpBlocks.upload( T.cpBlocks.data(), T.cpBlocks.size()*2); c->dCpPages.upload( T.cpPages.data(), T.cpPages.size()); }
		if (!T.nullable.empty()) c->dNullable.upload( T.nullable.data(), T.nullable.size()*sizeof(DevNullable));
		if (!T.approx.empty()) c->dApprox.upload( T.approx.data(), T.approx.size()*sizeof(DevApproxPattern));
		c->dShapes.upload( T.shapes.data(), T.shapes.size()*sizeof(DevShape));
		c->dShapePats.upload( T.shapePats.data(), T.shapePats.size()*4);
		// literals and word shapes by the words kernel: plain tables (words by ASCII word characters, no classes by code point, no empty matches)
		c->wordsKernel = T.approx.empty() && !T.ucp && T.cpBlocks.empty() && T.nullable.empty() && !getenv( "SPA_L1_NO_WORDS_KERNEL");
		if (!c->wordsKernel && T.nofShapes) throw std::runtime_error( "internal: word shapes in a table the words kernel does not take");
		{
			// image of all passes: what the kernels that walk an automaton backwards read (global memory)
			std::vector<uint64_t> img;
			img.insert( img.end(), T.charMask.begin(), T.charMask.end());
			c->imgAccept = (uint32_t)img.size(); img.insert( img.end(), T.acceptMask.begin(), T.acceptMask.end());
			c->imgStart = (uint32_t)img.size(); img.insert( img.end(), T.startMask.begin(), T.startMask.end());
			c->imgShift = (uint32_t)img.size(); img.insert( img.end(), T.shiftDst.begin(), T.shiftDst.end());
			c->imgSelf = (uint32_t)img.size(); img.insert( img.end(), T.selfLoop.begin(), T.selfLoop.end());
			c->imgExSrc = (uint32_t)img.size(); img.insert( img.end(), T.exSrc.begin(), T.exSrc.end());
			c->imgExDst = (uint32_t)img.size(); img.insert( img.end(), T.exDst.begin(), T.exDst.end());
			c->imgShapeFp = (uint32_t)img.size(); img.insert( img.end(), T.shapeFp.begin(), T.shapeFp.end());	// (the compact shape table rides along: staged in LDS with the rest)
			c->dTableImage.upload( img.data(), img.size()*8);
			c->imgWords = (uint32_t)img.size();
		}
		if (c->wordsKernel)
		{
			// image of the words kernel: it walks the patterns of the passes BEHIND the scanned ones only, so those passes and the compact
			// shape table are all it stages in LDS (the 10k set: 101 KB instead of 124: room for 16 waves per workgroup).  The kernel
			// indexes by absolute pass: the offsets carry the bias (modulo 2^32).
			const size_t sp = T.scanPasses, mx = T.maxExceptions;
			auto tail = [&]( std::vector<uint64_t>& img, const std::vector<uint64_t>& v, size_t perPass) -> uint32_t
			{
				size_t skip = sp*perPass < v.size() ? sp*perPass : v.size();
				uint32_t off = (uint32_t)img.size() - (uint32_t)skip;
				img.insert( img.end(), v.begin() + skip, v.end());
				return off;
			};
			std::vector<uint64_t> img;
			c->wChar = tail( img, T.charMask, (size_t)T.nofClasses*64);
			c->wAccept = tail( img, T.acceptMask, (size_t)CTX_COUNT*64);
			c->wStart = tail( img, T.startMask, (size_t)CTX_COUNT*64);
			c->wShift = tail( img, T.shiftDst, 64);
			c->wSelf = tail( img, T.selfLoop, 64);
			c->wExSrc = tail( img, T.exSrc, mx*64);
			c->wExDst = tail( img, T.exDst, mx*64);
			c->wShapeFp = (uint32_t)img.size(); img.insert( img.end(), T.shapeFp.begin(), T.shapeFp.end());
			if (img.empty()) img.push_back( 0);
			c->dWordsImage.upload( img.data(), img.size()*8);
			c->wWords = (uint32_t)img.size();
		}
		{
			// LDS image of the scan kernel: the passes it runs (the word shapes' passes behind them are never scanned), when it fits
			// (one copy per workgroup; bigger workgroups when the copy is big)
			const uint32_t sp = c->wordsKernel ? T.scanPasses : T.nofPasses;
			const size_t maxEx = T.maxExceptions ? T.maxExceptions : 1;
			std::vector<uint64_t> img;
			img.insert( img.end(), T.charMask.begin(), T.charMask.begin() + (size_t)sp*T.nofClasses*64);
			c->ldsAccept = (uint32_t)img.size(); img.insert( img.end(), T.acceptMask.begin(), T.acceptMask.begin() + (size_t)sp*CTX_COUNT*64);
			c->ldsStart = (uint32_t)img.size(); img.insert( img.end(), T.startMask.begin(), T.startMask.begin() + (size_t)sp*CTX_COUNT*64);
			c->ldsShift = (uint32_t)img.size(); img.insert( img.end(), T.shiftDst.begin(), T.shiftDst.begin() + (size_t)sp*64);
			c->ldsSelf = (uint32_t)img.size(); img.insert( img.end(), T.selfLoop.begin(), T.selfLoop.begin() + (size_t)sp*64);
			c->ldsExSrc = (uint32_t)img.size(); img.insert( img.end(), T.exSrc.begin(), T.exSrc.begin() + (size_t)sp*maxEx*64);
			c->ldsExDst = (uint32_t)img.size(); img.insert( img.end(), T.exDst.begin(), T.exDst.begin() + (size_t)sp*maxEx*64);
			if (img.empty()) img.push_back( 0);
			size_t bytes = img.size()*8;
			c->dScanImage.upload( img.data(), bytes);
			if (bytes <= 144*1024 && sp <= 8)
			{
				c->ldsWords = (uint32_t)img.size();
				// as many workgroups per CU as copies of the image fit into the 160 KB of LDS, sharing the waves
				// the register budget allows (5 per SIMD up to 2 passes, 4 beyond; two 10-wave workgroups of the
				// 3-pass instance at 96 registers were measured not to share a CU)
				const unsigned maxWaves = sp <= 2 ? 20u : 16u;
				unsigned maxCopies = (unsigned)((160*1024 - 1024) / (bytes ? bytes : 1));
				if (maxCopies < 1) maxCopies = 1;
				if (maxCopies > 5) maxCopies = 5;
				// waves per workgroup in multiples of 4 (one per SIMD): workgroups of 6 or 10 waves load the four
				// SIMDs unevenly and the next workgroup does not fit beside them (measured: 3 x 6 waves of the
				// 2-pass instance ran at the speed of 10-12 resident waves)
				unsigned best = 0, wpb = 4;
				for (unsigned cp=maxCopies; cp>=1; --cp)
				{
					unsigned wv = (maxWaves / cp) & ~3u;
					if (wv > 16) wv = 16;
					if (wv * cp > best) { best = wv * cp; wpb = wv; }
				}
				c->blockThreads = 64 * wpb;
			}
			else { c->ldsWords = 0; c->blockThreads = 256; }
		}
		c->dCounters.alloc( L1C_ALLOC*sizeof(uint64_t));
		uint32_t npat = (uint32_t)T.patterns.size();
		c->queueCap = 4096 > 2*npat+256 ? 4096 : 2*npat+256;
		HIP_CHECK( hipStreamCreateWithFlags( &c->own, hipStreamNonBlocking));
		HIP_CHECK( hipEventCreate( &c->evStart));
		HIP_CHECK( hipEventCreate( &c->evMid));
		HIP_CHECK( hipEventCreate( &c->evWords));
		HIP_CHECK( hipEventCreate( &c->evStop));
		return c;
	}
	catch (const std::exception& e)
	{
		l->lasterror = e.what();
		delete c;
		return 0;
	}
}

void sp_lexer_ctx_free( sp_lexer_ctx_t* c)
{
	if (!c) return;
	if (c->own) { (void)hipSetDevice( c->device); (void)hipStreamSynchronize( c->own); (void)hipStreamDestroy( c->own); }
	if (c->evStart) (void)hipEventDestroy( c->evStart);
	if (c->evMid) (void)hipEventDestroy( c->evMid);
	if (c->evWords) (void)hipEventDestroy( c->evWords);
	if (c->evStop) (void)hipEventDestroy( c->evStop);
	delete c;
}
const char* sp_lexer_ctx_last_error( const sp_lexer_ctx_t* c) { return c->lasterror.c_str(); }
int sp_lexer_ctx_reset( sp_lexer_ctx_t*) { return SP_OK; }	// the context keeps no per-document state between calls

int sp_lexer_ctx_reserve_output( sp_lexer_ctx_t* c, uint64_t lexems)
{
	if (lexems > c->minLexemCapacity) c->minLexemCapacity = lexems;
	return SP_OK;
}
int sp_lexer_ctx_grow_arena( sp_lexer_ctx_t* c)
{
	// SP_DOC_ERR_ARENA has two sources: the slice of a report queue of some scan unit (queueMul), or the event array of some
	// document (eventCap).  The last batch counted them apart; only what overflowed doubles (both when nothing is known).
	bool queue = true, events = true;
	if (c->evValid && hipSetDevice( c->device) == hipSuccess && hipStreamSynchronize( c->lastStream) == hipSuccess)
	{
		uint64_t over[ 2] = {0, 0};
		if (hipMemcpyAsync( over, (const uint64_t*)c->dCounters.ptr + L1C_OVER_QUEUE, sizeof(over), hipMemcpyDeviceToHost, c->own) == hipSuccess
		&&  hipStreamSynchronize( c->own) == hipSuccess && (over[ 0] || over[ 1])) { queue = over[ 0] != 0; events = over[ 1] != 0; }
	}
	if (events)
	{
		if (c->eventCap >= (1u<<26)) { c->lasterror = "arena at its maximum size"; return SP_ERR_INVALID; }
		c->eventCap *= 2; c->arenaWaves = 0;
	}
	if (queue)
	{
		if (c->queueMul >= 4096) { c->lasterror = "report queue at its maximum size (4096 x 1/16 reports per text byte)"; return SP_ERR_INVALID; }
		c->queueCap *= 2; c->queueMul *= 2;
	}
	return SP_OK;
}

} // extern "C"

namespace {
// a copy on the context's own stream, complete when the call returns
inline void copySync( sp_lexer_ctx* c, void* dst, const void* src, size_t n, hipMemcpyKind kind)
{
	HIP_CHECK( hipMemcpyAsync( dst, src, n, kind, c->own));
	HIP_CHECK( hipStreamSynchronize( c->own));
}
enum {SPA_L1_POST_WAVES_PER_EU_DEFAULT=6};
void launchLex( sp_lexer_ctx* c, const void* d_text, const void* d_doc_offsets, size_t ndocs, size_t nbytes, hipStream_t stream)
{
	HIP_CHECK( hipSetDevice( c->device));
	const LexTables& T = c->inst->compiler.tables();
	// documents longer than a chunk are scanned as several units (SPA_L1_CHUNK_BYTES: tests)
	uint32_t chunkBytes = 32768;		// (12288 x 64 KiB documents: scan 102.4 ms unchunked, 95.9 / 95.5 / 97.0 ms at 32 / 16 / 4 KiB chunks)
	// (an expression that can stay live across blanks -- <[^>]*>, ".*" with DOTALL -- fails the warm-up proof of nearly every chunk:
	//  such tables are scanned document by document, the chunked pass would only be thrown away)
	if (!T.lanesOk) chunkBytes = 0xFFFFFFC0u;
	if (const char* e = getenv( "SPA_L1_CHUNK_BYTES")) { long v = atol( e); if (v >= 64 && v <= (1l << 30)) chunkBytes = (uint32_t)v & ~63u; }
	const uint64_t maxUnits = (uint64_t)ndocs + (uint64_t)nbytes / chunkBytes + 2;
	if (maxUnits >= 0xFFFFFFFFull) throw std::runtime_error( "too many scan units in one batch");
	// scan kernel: one wave per unit up to what the device holds (a wave without a unit leaves at once)
	unsigned wavesWanted = (unsigned)((maxUnits < (uint64_t)c->numCUs*20) ? maxUnits : (uint64_t)c->numCUs*20);
	const unsigned wpb = c->blockThreads / 64;
	unsigned nblocks = (wavesWanted + wpb-1) / wpb;
	if (nblocks == 0) nblocks = 1;
	// the post-processing kernel (and the approximate-matching kernel) has its own number of waves: one event array each
	unsigned postPerCU = 4*SPA_L1_POST_WAVES_PER_EU_DEFAULT;	// (matches the register budget of the kernel, l1_kernel.hip)
	if (const char* e = getenv( "SPA_L1_POST_WAVES_PER_CU")) { int v = atoi( e); if (v >= 1 && v <= 40) postPerCU = (unsigned)v; }
	unsigned nwaves = (unsigned)((ndocs < (size_t)c->numCUs*postPerCU) ? ndocs : (size_t)c->numCUs*postPerCU);
	nwaves = (nwaves + 3u) & ~3u;
	if (nwaves == 0) nwaves = 4;
	uint64_t perWaveWords = 4ull*c->eventCap;		// the handler's event array (the report queue is per document: dQueue)
	{
		size_t maxWaves = ((size_t)48 << 30) / (perWaveWords*4);
		if (maxWaves < 4) maxWaves = 4;
		if (nwaves > maxWaves) nwaves = (unsigned)(maxWaves & ~(size_t)3);
	}
	if (c->arenaWaves < nwaves || c->arenaWords != perWaveWords)
	{
		size_t full = (size_t)c->numCUs*postPerCU;
		if (full * perWaveWords*4 > ((size_t)48 << 30)) full = ((size_t)48 << 30) / (perWaveWords*4);
		unsigned alloc = (nwaves >= 64 && nwaves < full) ? (unsigned)full : nwaves;		// (single documents, the plugin path: a small arena per context)
		c->arenaWaves = 0;
		c->dArena.alloc( (size_t)alloc * perWaveWords * 4);
		c->arenaWaves = alloc; c->arenaWords = perWaveWords;
	}
	uint64_t want = (uint64_t)nbytes/3 + 4096;
	if (want < c->minLexemCapacity) want = c->minLexemCapacity;
	if (c->lexemCapacity < want) { c->lexemCapacity = 0; c->dLexems.alloc( want*sizeof(sp_lexem_t)); c->lexemCapacity = want; }	// (capacity follows the buffer also when the allocation fails)
	c->dDocRange.reserve( (ndocs+1)*2*sizeof(uint64_t));
	c->dDocStatus.reserve( (ndocs+1)*sizeof(int32_t));
	c->dReportCount.reserve( (maxUnits+1)*sizeof(uint32_t));
	c->dUnitStart.reserve( (ndocs+2)*sizeof(uint32_t));
	c->dDocSequential.reserve( (ndocs+1)*sizeof(uint32_t));
	{
		// the report queues follow the text size: refuse a batch whose queues would not fit beside it instead of running into hipMalloc
		const uint64_t qbytes = ((((uint64_t)nbytes * c->queueMul) >> 4) + 64ull*(maxUnits+2)) * 16;
		size_t freeB = 0, totalB = 0;
		if (hipMemGetInfo( &freeB, &totalB) == hipSuccess)
		{
			const uint64_t have = (uint64_t)c->dQueue.bytes + (c->wordsKernel ? (uint64_t)c->dWordQueue.bytes : 0);
			const uint64_t want = qbytes * (c->wordsKernel ? 2 : 1);
			if (want > have && want - have > (uint64_t)(0.8 * (double)freeB))
			{
				char msg[ 200];
				snprintf( msg, sizeof(msg), "the report queues of this batch (%llu MB at %u/16 reports per text byte) do not fit the free device memory (%llu MB): pass fewer bytes per batch",
					(unsigned long long)(want >> 20), c->queueMul, (unsigned long long)(freeB >> 20));
				throw std::runtime_error( msg);
			}
		}
	}
	c->dQueue.reserve( ((((uint64_t)nbytes * c->queueMul) >> 4) + 64ull*(maxUnits+2)) * 16);
	if (c->wordsKernel)
	{
		c->dWordQueue.reserve( ((((uint64_t)nbytes * c->queueMul) >> 4) + 64ull*(maxUnits+2)) * 16);
		c->dWordCount.reserve( (maxUnits+1)*sizeof(uint32_t));
	}
	if (!T.approx.empty())
	{
		// approximate literal table: the decoded characters of every document (code point, byte offset)
		c->dCharCp.reserve( ((uint64_t)nbytes + ndocs + 64) * 4);
		c->dCharPos.reserve( ((uint64_t)nbytes + ndocs + 64) * 4);
	}
	HIP_CHECK( hipMemsetAsync( c->dCounters.ptr, 0, L1C_ALLOC*sizeof(uint64_t), stream));

	L1Params P;
	std::memset( &P, 0, sizeof(P));
	P.byteClass = (const uint8_t*)c->dByteClass.ptr; P.classCtx = (const uint8_t*)c->dClassCtx.ptr;
	P.charMask = (const uint64_t*)c->dCharMask.ptr; P.startMask = (const uint64_t*)c->dStartMask.ptr;
	P.acceptMask = (const uint64_t*)c->dAcceptMask.ptr; P.shiftDst = (const uint64_t*)c->dShiftDst.ptr;
	P.selfLoop = (const uint64_t*)c->dSelfLoop.ptr; P.exSrc = (const uint64_t*)c->dExSrc.ptr; P.exDst = (const uint64_t*)c->dExDst.ptr;
	P.exCount = (const uint32_t*)c->dExCount.ptr;
	P.patOfBit = (const uint32_t*)c->dPatOfBit.ptr; P.patterns = (const DevLexPattern*)c->dPatterns.ptr;
	P.symbols = (const DevSymbol*)c->dSymbols.ptr; P.symbolText = (const uint8_t*)c->dSymbolText.ptr;
	P.symbolMask = (uint32_t)T.symbols.size()-1;
	P.literals = (const DevLiteral*)c->dLiterals.ptr; P.literalText = (const uint8_t*)c->dLiteralText.ptr;
	P.litPats = (const uint32_t*)c->dLitPats.ptr; P.literalMask = (uint32_t)T.literals.size()-1; P.nofLiterals = T.nofLiterals;
	P.reportsOrdered = T.reportsOrdered ? 1u : 0u;
	P.nofPasses = T.nofPasses; P.nofClasses = T.nofClasses; P.maxExceptions = T.maxExceptions ? T.maxExceptions : 1;
	P.nofPatterns = (uint32_t)T.patterns.size();
	P.text = (const uint8_t*)d_text; P.docOffsets = (const uint64_t*)d_doc_offsets; P.ndocs = (uint32_t)ndocs;
	P.arenaBase = (uint32_t*)c->dArena.ptr; P.arenaWords = perWaveWords; P.queueCap = c->queueCap; P.eventCap = c->eventCap;
	P.counters = (uint64_t*)c->dCounters.ptr; P.lexems = (uint32_t*)c->dLexems.ptr; P.lexemCapacity = c->lexemCapacity;
	P.docRange = (uint64_t*)c->dDocRange.ptr; P.docStatus = (int32_t*)c->dDocStatus.ptr;
	P.reportQueue = (uint32_t*)c->dQueue.ptr; P.reportCount = (uint32_t*)c->dReportCount.ptr; P.queueMul = c->queueMul;
	P.approx = T.approx.empty() ? 0 : (const DevApproxPattern*)c->dApprox.ptr; P.nofApprox = (uint32_t)T.approx.size();
	P.charCp = (uint32_t*)c->dCharCp.ptr; P.charPos = (uint32_t*)c->dCharPos.ptr;
	P.cpBlocks = T.cpBlocks.empty() ? 0 : (const uint16_t*)c->dCpBlocks.ptr; P.cpPages = (const uint8_t*)c->dCpPages.ptr;
	P.ucp = T.ucp ? 1u : 0u;
	P.nullable = T.nullable.empty() ? 0 : (const DevNullable*)c->dNullable.ptr; P.nofNullable = (uint32_t)T.nullable.size();
	P.unitStart = (uint32_t*)c->dUnitStart.ptr; P.chunkBytes = chunkBytes; P.docSequential = (uint32_t*)c->dDocSequential.ptr; P.sequentialPass = 0;
	P.splitPatterns = (T.patterns.size() != c->inst->compiler.nofDefinitions()) ? 1u : 0u;
	P.shapes = (const DevShape*)c->dShapes.ptr; P.shapePats = (const uint32_t*)c->dShapePats.ptr; P.shapeMask = (uint32_t)T.shapes.size()-1;
	P.nofShapeVariants = (uint32_t)T.shapeVariants.size();
	for (size_t i=0; i<T.shapeVariants.size() && i<SHAPE_MAXVARIANTS; ++i) P.shapeVariants[ i] = T.shapeVariants[ i];
	P.shapeFpOffset = c->imgShapeFp; P.shapeSalt = T.shapeSalt;
	P.wordQueue = (uint32_t*)c->dWordQueue.ptr; P.wordCount = (uint32_t*)c->dWordCount.ptr; P.wordsKernel = c->wordsKernel ? 1u : 0u;
	HIP_CHECK( hipEventRecord( c->evStart, stream));
	// the post-processing kernel, which walks the scanned patterns backwards, reads the image of all passes from global memory ...
	P.tableImage = (const uint64_t*)c->dTableImage.ptr; P.ldsWords = 0;
	P.ldsChar = 0; P.ldsAccept = c->imgAccept; P.ldsStart = c->imgStart; P.ldsShift = c->imgShift; P.ldsSelf = c->imgSelf;
	P.ldsExSrc = c->imgExSrc; P.ldsExDst = c->imgExDst;
	// ... the scan kernel stages the image of the passes it runs in LDS
	L1Params PS = P;
	PS.nofPasses = c->wordsKernel ? T.scanPasses : T.nofPasses;
	PS.tableImage = (const uint64_t*)c->dScanImage.ptr; PS.ldsWords = c->ldsWords;
	PS.ldsAccept = c->ldsAccept; PS.ldsStart = c->ldsStart; PS.ldsShift = c->ldsShift; PS.ldsSelf = c->ldsSelf;
	PS.ldsExSrc = c->ldsExSrc; PS.ldsExDst = c->ldsExDst;
	if (PS.nofPasses == 0) HIP_CHECK( hipMemsetAsync( c->dReportCount.ptr, 0, (maxUnits+1)*sizeof(uint32_t), stream));
	// words kernel: a wave per unit, workgroups of 16 (12) waves that share one LDS copy of ITS image -- the passes behind the scanned ones + the shape table -- when it fits
	L1Params PW = P;
	// (16 waves per workgroup while the image leaves room for their rings and run ends, else 12; SPA_L1_WORD_WAVES=12: A/B runs)
	unsigned wordWaves = ((size_t)c->wWords*8 + (size_t)L1_WORD_WAVES_SMALL*L1_WORDS_LDS_PER_WAVE <= 160*1024 && !getenv( "SPA_L1_WORD_WAVES")) ? (unsigned)L1_WORD_WAVES_SMALL : (unsigned)L1_WORD_WAVES;
	if (c->wordsKernel)
	{
		PW.tableImage = (const uint64_t*)c->dWordsImage.ptr;
		PW.ldsChar = c->wChar; PW.ldsAccept = c->wAccept; PW.ldsStart = c->wStart; PW.ldsShift = c->wShift; PW.ldsSelf = c->wSelf;
		PW.ldsExSrc = c->wExSrc; PW.ldsExDst = c->wExDst; PW.shapeFpOffset = c->wShapeFp;
	}
	PW.ldsWords = ((size_t)c->wWords*8 + (size_t)wordWaves*L1_WORDS_LDS_PER_WAVE <= 160*1024 && T.nofShapes) ? c->wWords : 0u;
	unsigned wordBlocks = (unsigned)((maxUnits + wordWaves-1) / wordWaves < (uint64_t)c->numCUs ? (maxUnits + wordWaves-1) / wordWaves : (uint64_t)c->numCUs);
	if (wordBlocks == 0) wordBlocks = 1;
	// (scanWords = 0 keeps the batch off the lane-per-stream scan kernel: an expression that can stay live across blanks would
	//  fail the warm-up proof of most pieces, SPA_L1_NO_LANES: tests)
	P.postClusters = getenv( "SPA_L1_POST_SEQ") ? 0u : 1u;		// (SPA_L1_POST_SEQ: tests and A/B runs, one report after the other)
	P.scanWords = PS.scanWords = PW.scanWords = (c->wordsKernel && T.lanesOk && !getenv( "SPA_L1_NO_LANES")) ? T.scanWords : 0u;
	// lane-per-stream scan kernel (a few automaton words left to scan): a wave per unit, workgroups of four waves
	unsigned laneBlocks = (unsigned)((maxUnits + 3) / 4 < (uint64_t)c->numCUs*4 ? (maxUnits + 3) / 4 : (uint64_t)c->numCUs*4);
	if (laneBlocks == 0) laneBlocks = 1;
	HIP_CHECK( launchL1Lex( PS, PW, P, nblocks, c->blockThreads, laneBlocks, wordBlocks, wordWaves, nwaves, stream, c->evMid, c->evWords));
	c->wordsKernelName = !c->wordsKernel || P.nofApprox ? "(none)" : (wordWaves == (unsigned)L1_WORD_WAVES_SMALL ? "spa_l1_words_kernel_w16" : "spa_l1_words_kernel");
	if (P.nofApprox) std::snprintf( c->scanKernel, sizeof(c->scanKernel), "spa_l1_approx_kernel");
	else if (l1ScanByLanes( PS, P)) std::snprintf( c->scanKernel, sizeof(c->scanKernel), "spa_l1_scan_lanes_kernel");
	else if (PS.nofPasses == 0) std::snprintf( c->scanKernel, sizeof(c->scanKernel), "(none)");
	else std::snprintf( c->scanKernel, sizeof(c->scanKernel), "spa_l1_scan_kernel_p%u", PS.nofPasses <= 8 ? PS.nofPasses : PS.nofPasses <= 16 ? 16u : 32u);
	HIP_CHECK( hipEventRecord( c->evStop, stream));
	c->evValid = true; c->lastStream = stream; c->lastNdocs = ndocs;
}
}

extern "C" {

int sp_lexer_ctx_match_docs_device( sp_lexer_ctx_t* c, const void* d_text, const void* d_doc_offsets,
				    size_t ndocs, size_t nbytes, void* stream, sp_lex_device_batch_t* out)
{
	return guardedCall1( c->lasterror, SP_ERR_INVALID, [&]{
		if (ndocs >= 0xFFFFFFFFull) throw std::runtime_error( "too many documents in one batch");
		launchLex( c, d_text, d_doc_offsets, ndocs, nbytes, (hipStream_t)stream);
		if (out)
		{
			out->ndocs = ndocs; out->d_lexems = c->dLexems.ptr; out->d_doc_ranges = c->dDocRange.ptr;
			out->d_doc_status = c->dDocStatus.ptr; out->d_counters = c->dCounters.ptr;
		}
	});
}

int sp_lexer_ctx_batch_counters( sp_lexer_ctx_t* c, uint64_t counters[8])
{
	return guardedCall1( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		uint64_t all[ L1C_ALLOC];
		copySync( c, all, c->dCounters.ptr, L1C_ALLOC*sizeof(uint64_t), hipMemcpyDeviceToHost);
		for (int i=0; i<L1C_COUNT; ++i) counters[ i] = all[ i];
#ifndef SPA_PROF
		// (the phase profile of a PROF build lives in 4..7) scan units of the batch and documents scanned again in one piece
		counters[ 4] = (uint32_t)all[ L1C_UNITS]; counters[ 5] = all[ L1C_SEQDOCS]; counters[ 6] = all[ L1C_WORDREPORTS];
#endif
	});
}

int sp_lexer_ctx_batch_status( sp_lexer_ctx_t* c, int32_t* status, size_t ndocs)
{
	return guardedCall1( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		if (ndocs > c->lastNdocs) ndocs = c->lastNdocs;
		if (ndocs) copySync( c, status, c->dDocStatus.ptr, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
	});
}

double sp_lexer_ctx_last_kernel_ms( sp_lexer_ctx_t* c)
{
	if (!c->evValid) return -1.0;
	float ms = 0.0f;
	if (hipEventSynchronize( c->evStop) != hipSuccess) return -1.0;
	if (hipEventElapsedTime( &ms, c->evStart, c->evStop) != hipSuccess) return -1.0;
	return (double)ms;
}

int sp_lexer_ctx_last_kernel_ms_split( sp_lexer_ctx_t* c, double* scan_ms, double* post_ms)
{
	*scan_ms = -1.0; *post_ms = -1.0;
	if (!c->evValid) return SP_ERR_INVALID;
	float a = 0.0f, b = 0.0f;
	if (hipEventSynchronize( c->evStop) != hipSuccess) return SP_ERR_INVALID;
	if (hipEventElapsedTime( &a, c->evStart, c->evMid) != hipSuccess) return SP_ERR_INVALID;
	if (hipEventElapsedTime( &b, c->evMid, c->evStop) != hipSuccess) return SP_ERR_INVALID;
	*scan_ms = (double)a; *post_ms = (double)b;
	return SP_OK;
}

// the scan kernel the last launch went through (bench.py prices and names the kernel that ran)
const char* sp_lexer_ctx_scan_kernel_name( const sp_lexer_ctx_t* c)
{
	return c->scanKernel;
}

const char* sp_lexer_ctx_words_kernel_name( const sp_lexer_ctx_t* c)
{
	return c->wordsKernelName;
}

// the same with the words kernel on its own (round 3: automaton scan | literals + word shapes | start of match + handler + ordinal positions)
int sp_lexer_ctx_last_kernel_ms_split3( sp_lexer_ctx_t* c, double* scan_ms, double* words_ms, double* post_ms)
{
	*scan_ms = -1.0; *words_ms = -1.0; *post_ms = -1.0;
	if (!c->evValid) return SP_ERR_INVALID;
	float a = 0.0f, b = 0.0f, d = 0.0f;
	if (hipEventSynchronize( c->evStop) != hipSuccess) return SP_ERR_INVALID;
	if (hipEventElapsedTime( &a, c->evStart, c->evMid) != hipSuccess) return SP_ERR_INVALID;
	if (hipEventElapsedTime( &b, c->evMid, c->evWords) != hipSuccess) return SP_ERR_INVALID;
	if (hipEventElapsedTime( &d, c->evWords, c->evStop) != hipSuccess) return SP_ERR_INVALID;
	*scan_ms = (double)a; *words_ms = (double)b; *post_ms = (double)d;
	return SP_OK;
}

int sp_lexer_ctx_match_docs( sp_lexer_ctx_t* c, const char* text, const uint64_t* doc_offsets, size_t ndocs, sp_lex_batch_t* out)
{
	std::memset( out, 0, sizeof(*out));
	int rc = guardedCall1( c->lasterror, SP_ERR_INVALID, [&]{
		if (ndocs >= 0xFFFFFFFFull) throw std::runtime_error( "too many documents in one batch");
		HIP_CHECK( hipSetDevice( c->device));
		size_t nbytes = ndocs ? (size_t)doc_offsets[ ndocs] : 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			if (doc_offsets[ di+1] - doc_offsets[ di] >= 0xFFFFFFFFull) throw std::runtime_error( "size of string to scan out of range");	// :866-869
		}
		c->dText.reserve( nbytes+16);
		c->dDocOffsets.reserve( (ndocs+1)*sizeof(uint64_t));
		if (nbytes) copySync( c, c->dText.ptr, text, nbytes, hipMemcpyHostToDevice);
		copySync( c, c->dDocOffsets.ptr, doc_offsets, (ndocs+1)*sizeof(uint64_t), hipMemcpyHostToDevice);
		uint64_t counters[ L1C_COUNT];
		std::vector<int32_t> st( ndocs+1);
		for (int attempt=0;; ++attempt)
		{
			launchLex( c, c->dText.ptr, c->dDocOffsets.ptr, ndocs, nbytes, c->own);
			HIP_CHECK( hipStreamSynchronize( c->own));
			copySync( c, counters, c->dCounters.ptr, sizeof(counters), hipMemcpyDeviceToHost);
			if (ndocs) copySync( c, st.data(), c->dDocStatus.ptr, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
			bool grow = false;
			if (counters[ L1C_LEXEMS] > c->lexemCapacity) { c->minLexemCapacity = counters[ L1C_LEXEMS] + counters[ L1C_LEXEMS]/8 + 1024; grow = true; }
			if (counters[ L1C_FAILED])
			{
				bool arena = false;
				for (size_t di=0; di<ndocs && !arena; ++di) arena = (st[ di] == SP_DOC_ERR_ARENA);
				if (arena && sp_lexer_ctx_grow_arena( c) == SP_OK) grow = true;
			}
			if (!grow || attempt >= 12) break;
		}
		std::vector<uint64_t> range( ndocs*2+2);
		if (ndocs) copySync( c, range.data(), c->dDocRange.ptr, ndocs*2*sizeof(uint64_t), hipMemcpyDeviceToHost);
		uint64_t nlex = counters[ L1C_LEXEMS] < c->lexemCapacity ? counters[ L1C_LEXEMS] : c->lexemCapacity;
		std::vector<sp_lexem_t> raw( nlex+1);
		if (nlex) copySync( c, raw.data(), c->dLexems.ptr, nlex*sizeof(sp_lexem_t), hipMemcpyDeviceToHost);
		out->ndocs = ndocs;
		out->doc_lexem_offsets = (uint64_t*)std::malloc( (ndocs+1)*sizeof(uint64_t));
		out->doc_status = (int32_t*)std::malloc( (ndocs+1)*sizeof(int32_t));
		uint64_t total = 0;
		for (size_t di=0; di<ndocs; ++di) { if (st[ di] != 0) range[ 2*di+1] = 0; total += range[ 2*di+1]; }
		out->lexems = (sp_lexem_t*)std::malloc( (total+1)*sizeof(sp_lexem_t));
		if (!out->doc_lexem_offsets || !out->doc_status || !out->lexems) throw std::bad_alloc();
		uint64_t lp = 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			out->doc_lexem_offsets[ di] = lp;
			out->doc_status[ di] = st[ di];
			if (range[ 2*di+1]) std::memcpy( out->lexems + lp, raw.data() + range[ 2*di], range[ 2*di+1]*sizeof(sp_lexem_t));
			lp += range[ 2*di+1];
		}
		out->doc_lexem_offsets[ ndocs] = lp;
		out->nlexems = lp;
		if (counters[ L1C_FAILED])
		{
			size_t bad = 0;
			while (bad < ndocs && st[ bad] == 0) ++bad;
			char msg[ 160];
			snprintf( msg, sizeof(msg), "at least one document failed: document %zu has status %d%s", bad, bad < ndocs ? st[ bad] : -1,
				(bad < ndocs && st[ bad] == SP_DOC_ERR_LEXEMSIZE) ? " (size of matched term out of range)" : "");
			throw std::runtime_error( msg);
		}
	});
	if (rc == SP_OK) return SP_OK;
	return c->lasterror.find( "document failed") != std::string::npos ? SP_ERR_MATCH : rc;
}

// host copy of the lexems of the documents [first_doc, first_doc+ndocs) of the last device batch
int sp_lexer_ctx_batch_fetch_docs( sp_lexer_ctx_t* c, size_t first_doc, size_t ndocs, sp_lex_batch_t* out)
{
	std::memset( out, 0, sizeof(*out));
	return guardedCall1( c->lasterror, SP_ERR_DEVICE, [&]{
		HIP_CHECK( hipSetDevice( c->device));
		HIP_CHECK( hipStreamSynchronize( c->lastStream));
		if (first_doc > c->lastNdocs || ndocs > c->lastNdocs - first_doc) throw std::runtime_error( "document range outside the last batch");
		uint64_t counters[ L1C_COUNT];
		copySync( c, counters, c->dCounters.ptr, sizeof(counters), hipMemcpyDeviceToHost);
		const uint64_t devLexems = counters[ L1C_LEXEMS] < c->lexemCapacity ? counters[ L1C_LEXEMS] : c->lexemCapacity;
		std::vector<uint64_t> range( ndocs*2+2);
		out->ndocs = ndocs;
		out->doc_lexem_offsets = (uint64_t*)std::malloc( (ndocs+1)*sizeof(uint64_t));
		out->doc_status = (int32_t*)std::malloc( (ndocs+1)*sizeof(int32_t));
		if (!out->doc_lexem_offsets || !out->doc_status) throw std::bad_alloc();
		if (ndocs)
		{
			copySync( c, range.data(), (const uint64_t*)c->dDocRange.ptr + 2*first_doc, ndocs*2*sizeof(uint64_t), hipMemcpyDeviceToHost);
			copySync( c, out->doc_status, (const int32_t*)c->dDocStatus.ptr + first_doc, ndocs*sizeof(int32_t), hipMemcpyDeviceToHost);
		}
		uint64_t total = 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			if (out->doc_status[ di] != 0 || range[ 2*di] + range[ 2*di+1] > devLexems) range[ 2*di+1] = 0;
			total += range[ 2*di+1];
		}
		out->lexems = (sp_lexem_t*)std::malloc( (total+1)*sizeof(sp_lexem_t));
		if (!out->lexems) throw std::bad_alloc();
		uint64_t lp = 0;
		for (size_t di=0; di<ndocs; ++di)
		{
			out->doc_lexem_offsets[ di] = lp;
			if (range[ 2*di+1]) copySync( c, out->lexems + lp, (const sp_lexem_t*)c->dLexems.ptr + range[ 2*di], range[ 2*di+1]*sizeof(sp_lexem_t), hipMemcpyDeviceToHost);
			lp += range[ 2*di+1];
		}
		out->doc_lexem_offsets[ ndocs] = lp;
		out->nlexems = lp;
	});
}

void sp_lex_batch_free( sp_lex_batch_t* b)
{
	std::free( b->lexems); std::free( b->doc_lexem_offsets); std::free( b->doc_status);
	std::memset( b, 0, sizeof(*b));
}

int sp_lexer_ctx_match( sp_lexer_ctx_t* c, const char* src, size_t srclen, sp_lexem_t** lexems, size_t* nlexems)
{
	uint64_t offs[2] = {0, (uint64_t)srclen};
	sp_lex_batch_t b;
	int rc = sp_lexer_ctx_match_docs( c, src, offs, 1, &b);
	if (rc != SP_OK) { sp_lex_batch_free( &b); *lexems = 0; *nlexems = 0; return rc; }
	*lexems = b.lexems; *nlexems = b.nlexems; b.lexems = 0;
	sp_lex_batch_free( &b);
	return SP_OK;
}

} // extern "C"
