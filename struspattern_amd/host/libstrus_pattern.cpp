// C++ host side of the drop-in: implements strus's PatternLexerInterface / PatternMatcherInterface
// families on top of the C-ABI (include/strus_pattern_amd.h -> HIP kernels) and exports the two
// factory functions the reference exports (include/strus/lib/pattern.hpp:27-32,
// src/libstrus_pattern.cpp:21-47).  Conventions kept from the reference:
//  * every factory / createInstance / createContext returns a new-allocated object owned by the caller
//  * no exception crosses the interface: errors go to ErrorBufferInterface::report(code, fmt, ...)
//    with the reference's context texts, the method returns void / 0 / false / empty vector
//  * an Instance is immutable after compile() and must outlive its Contexts
#include "strus/lib/pattern.hpp"
#include "strus/errorBufferInterface.hpp"
#include "strus/patternLexerInterface.hpp"
#include "strus/patternMatcherInterface.hpp"
#include "strus/lib/pattern_resultformat.hpp"
#include "../../include/strus_pattern_amd.h"
#include <atomic>
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace strus;

namespace {

// Which GPU a new context runs on.  The reference's threading model is one Context per thread over a shared Instance
// (tests/randomTokenPatternMatch/src/testRandomTokenPatternMatch.cpp:325-345); contexts are independent here too -- each has its own
// stream and buffers --, so they are dealt round-robin over the visible devices.  SPA_DEVICE=<n> pins every context to one device.
int nextDevice()
{
	static std::atomic<unsigned> counter( 0);
	const int ndev = sp_device_count();
	if (ndev <= 1) return 0;
	if (const char* e = std::getenv( "SPA_DEVICE"))
	{
		const int d = std::atoi( e);
		if (d >= 0 && d < ndev) return d;
	}
	return (int)(counter.fetch_add( 1) % (unsigned)ndev);
}

ErrorCode codeOf( int rc)
{
	switch (rc)
	{
		case SP_ERR_NOMEM: return ErrorCodeOutOfMem;
		case SP_ERR_COMPILE: return ErrorCodeSyntax;
		default: return ErrorCodeRuntimeError;
	}
}

// ---------------------------------------------------------------- lexer
class LexerContext :public PatternLexerContextInterface
{
public:
	LexerContext( sp_lexer_ctx_t* h, ErrorBufferInterface* e) :m_h(h),m_errorhnd(e){}
	virtual ~LexerContext() { sp_lexer_ctx_free( m_h); }
	virtual std::vector<analyzer::PatternLexem> match( const char* src, std::size_t srclen)
	{
		std::vector<analyzer::PatternLexem> rt;
		sp_lexem_t* lex = 0; size_t n = 0;
		int rc = sp_lexer_ctx_match( m_h, src, srclen, &lex, &n);
		if (rc != SP_OK)
		{
			m_errorhnd->report( codeOf( rc), "failed to run pattern matching terms with regular expressions: %s", sp_lexer_ctx_last_error( m_h));
			return rt;
		}
		rt.reserve( n);
		for (size_t i=0; i<n; ++i) rt.push_back( analyzer::PatternLexem( lex[i].id, lex[i].ordpos, analyzer::Position( 0/*origseg*/, (int)lex[i].origpos), lex[i].origsize));
		sp_free( lex);
		return rt;
	}
	virtual void reset() { sp_lexer_ctx_reset( m_h); }
private:
	sp_lexer_ctx_t* m_h;
	ErrorBufferInterface* m_errorhnd;
};

class LexerInstance :public PatternLexerInstanceInterface
{
public:
	explicit LexerInstance( ErrorBufferInterface* e) :m_h(sp_lexer_create()),m_errorhnd(e) { if (!m_h) throw std::bad_alloc(); }
	virtual ~LexerInstance() { sp_lexer_free( m_h); }

	virtual void defineLexemName( unsigned int id, const std::string& name_)
	{
		check( sp_lexer_define_lexem_name( m_h, id, name_.c_str()), "failed to assign lexem name to lexem or symbol identifier: %s");
	}
	virtual const char* getLexemName( unsigned int id) const { return sp_lexer_get_lexem_name( m_h, id); }
	virtual void defineLexem( unsigned int id, const std::string& expression, unsigned int resultIndex, unsigned int level, analyzer::PositionBind posbind)
	{
		check( sp_lexer_define_lexem( m_h, id, expression.c_str(), resultIndex, level, (int)posbind), "failed to define term match regular expression pattern: %s");
	}
	virtual void defineSymbol( unsigned int symbolid, unsigned int patternid, const std::string& name_)
	{
		check( sp_lexer_define_symbol( m_h, symbolid, patternid, name_.c_str()), "failed to define regular expression pattern symbol: %s");
	}
	virtual unsigned int getSymbol( unsigned int patternid, const std::string& name_) const { return sp_lexer_get_symbol( m_h, patternid, name_.c_str()); }
	virtual void defineOption( const std::string& name_, double value)
	{
		check( sp_lexer_define_option( m_h, name_.c_str(), value), "define option failed for hyperscan pattern lexer: %s");
	}
	virtual bool compile()
	{
		return check( sp_lexer_compile( m_h), "failed to compile regular expression patterns: %s");
	}
	virtual PatternLexerContextInterface* createContext() const
	{
		sp_lexer_ctx_t* c = sp_lexer_ctx_create( m_h, nextDevice());
		if (!c) { m_errorhnd->report( ErrorCodeRuntimeError, "failed to create term match context: %s", sp_lexer_last_error( m_h)); return 0; }
		try { return new LexerContext( c, m_errorhnd); }
		catch (const std::bad_alloc&) { sp_lexer_ctx_free( c); m_errorhnd->report( ErrorCodeOutOfMem, "memory allocation error in %s", "strus pattern"); return 0; }
	}
	virtual const char* name() const { return "std"; }
	virtual StructView view() const { return StructView()( "name", name()); }
private:
	bool check( int rc, const char* fmt) const
	{
		if (rc == SP_OK) return true;
		m_errorhnd->report( codeOf( rc), fmt, sp_lexer_last_error( m_h));
		return false;
	}
	sp_lexer_t* m_h;
	ErrorBufferInterface* m_errorhnd;
};

class Lexer :public PatternLexerInterface
{
public:
	explicit Lexer( ErrorBufferInterface* e) :m_errorhnd(e){}
	virtual std::vector<std::string> getCompileOptionNames() const
	{
		static const char* ar[] = {"CASELESS", "DOTALL", "MULTILINE", "ALLOWEMPTY", "UCP", 0};	// src/patternLexer.cpp:1154-1163
		std::vector<std::string> rt;
		for (int i=0; ar[i]; ++i) rt.push_back( ar[i]);
		return rt;
	}
	virtual PatternLexerInstanceInterface* createInstance() const
	{
		try { return new LexerInstance( m_errorhnd); }
		catch (const std::bad_alloc&) { m_errorhnd->report( ErrorCodeOutOfMem, "failed to create term match instance: %s", "out of memory"); return 0; }
	}
	virtual const char* name() const { return "std"; }
	virtual StructView view() const { return StructView()( "name", name())( "description", "Pattern lexer with a bit-parallel multi-regex automaton on AMD CDNA4 GPUs"); }
private:
	ErrorBufferInterface* m_errorhnd;
};

// ---------------------------------------------------------------- matcher
class MatcherContext :public PatternMatcherContextInterface
{
public:
	MatcherContext( sp_matcher_ctx_t* h, const sp_matcher_t* inst, const std::vector<const PatternResultFormat*>* formats, ErrorBufferInterface* e)
		:m_h(h),m_inst(inst),m_formats(formats),m_errorhnd(e),m_resultFormatContext(e){}
	virtual ~MatcherContext() { sp_matcher_ctx_free( m_h); }
	virtual void putInput( const analyzer::PatternLexem& term)
	{
		sp_lexem_t lx; lx.id = term.id(); lx.ordpos = term.ordpos(); lx.origpos = (uint32_t)term.origpos().ofs(); lx.origsize = (uint32_t)term.origsize();
		uint32_t seg = (uint32_t)term.origpos().seg();
		int rc = sp_matcher_ctx_put_input( m_h, &lx, seg ? &seg : 0, 1);
		if (rc != SP_OK) m_errorhnd->report( codeOf( rc), "failed to feed input to pattern matcher: %s", sp_matcher_ctx_last_error( m_h));
	}
	virtual std::vector<analyzer::PatternMatcherResult> fetchResults()
	{
		std::vector<analyzer::PatternMatcherResult> rt;
		sp_result_t* res = 0; sp_result_item_t* items = 0; size_t nres = 0, nitems = 0;
		int rc = sp_matcher_ctx_fetch_results( m_h, &res, &nres, &items, &nitems);
		if (rc != SP_OK)
		{
			m_errorhnd->report( codeOf( rc), "failed to fetch pattern match result: %s", sp_matcher_ctx_last_error( m_h));
			return rt;
		}
		const uint32_t* resfmt = 0; const uint32_t* itemfmt = 0;
		(void)sp_matcher_ctx_fetch_formats( m_h, &resfmt, &itemfmt);
		try
		{
			rt.reserve( nres);
			for (size_t ri=0; ri<nres; ++ri)
			{
				const sp_result_t& r = res[ri];
				// src/patternMatcher.cpp:248-269: a result with a format string gets a value built from
				// its items and an empty item list, any other result keeps its items
				std::vector<analyzer::PatternMatcherResultItem> il;
				gatherResultItems( il, items, itemfmt, r.item_begin, r.item_begin + r.item_count);
				const char* value = 0;
				const uint32_t fmt = resfmt ? resfmt[ ri] : 0;
				if (fmt)
				{
					value = m_resultFormatContext.map( formatOf( fmt), il.data(), il.size());
					il.clear();
				}
				rt.push_back( analyzer::PatternMatcherResult( sp_matcher_pattern_name( m_inst, r.handle), value, r.ordpos, r.ordend,
						analyzer::Position( (int)r.origseg, (int)r.origpos), analyzer::Position( (int)r.origendseg, (int)r.origend), il));
			}
		}
		catch (const std::exception& e)
		{
			m_errorhnd->report( ErrorCodeRuntimeError, "failed to fetch pattern match result: %s", e.what());
			rt.clear();
		}
		sp_free( res); sp_free( items);
		return rt;
	}
	virtual analyzer::PatternMatcherStatistics getStatistics() const
	{
		analyzer::PatternMatcherStatistics stats;
		sp_matcher_stats_t st;
		if (sp_matcher_ctx_statistics( m_h, &st) == SP_OK)
		{
			stats.define( "nofProgramsInstalled", st.nofProgramsInstalled);
			stats.define( "nofAltKeyProgramsInstalled", st.nofAltKeyProgramsInstalled);
			stats.define( "nofSignalsFired", st.nofSignalsFired);
			stats.define( "nofTriggersAvgActive", st.nofTriggersAvgActive);
		}
		return stats;
	}
	virtual void reset() { sp_matcher_ctx_reset( m_h); m_resultFormatContext.clear(); }
private:
	const PatternResultFormat* formatOf( uint32_t handle) const
	{
		if (!m_formats || !handle || handle > m_formats->size() || !(*m_formats)[ handle-1]) throw std::runtime_error( "result refers to an undefined format string");
		return (*m_formats)[ handle-1];
	}
	// src/patternMatcher.cpp:164-190 over the flattened records: an item with a format handle is followed
	// by the records of its format arguments (include/strus_pattern_amd.h, "result format strings")
	void gatherResultItems( std::vector<analyzer::PatternMatcherResultItem>& out, const sp_result_item_t* items, const uint32_t* itemfmt, size_t begin, size_t end)
	{
		for (size_t i=begin; i<end; ++i)
		{
			const sp_result_item_t& it = items[ i];
			const char* value = 0;
			const uint32_t fmt = itemfmt ? itemfmt[ 2*i] : 0;
			if (fmt)
			{
				const size_t nsub = itemfmt[ 2*i+1];
				std::vector<analyzer::PatternMatcherResultItem> args;
				gatherResultItems( args, items, itemfmt, i+1, i+1+nsub);
				value = m_resultFormatContext.map( formatOf( fmt), args.data(), args.size());
				i += nsub;
			}
			out.push_back( analyzer::PatternMatcherResultItem( sp_matcher_variable_name( m_inst, it.variable), value, it.ordpos, it.ordend,
					analyzer::Position( (int)it.origseg, (int)it.origpos), analyzer::Position( (int)it.origendseg, (int)it.origend)));
		}
	}
	sp_matcher_ctx_t* m_h;
	const sp_matcher_t* m_inst;
	const std::vector<const PatternResultFormat*>* m_formats;
	ErrorBufferInterface* m_errorhnd;
	PatternResultFormatContext m_resultFormatContext;
};

class MatcherInstance :public PatternMatcherInstanceInterface, public PatternResultFormatVariableMap
{
public:
	explicit MatcherInstance( ErrorBufferInterface* e) :m_h(sp_matcher_create()),m_errorhnd(e),m_resultFormatTable(this,e) { if (!m_h) throw std::bad_alloc(); }
	virtual ~MatcherInstance() { sp_matcher_free( m_h); }
	// PatternResultFormatVariableMap (src/patternMatcher.cpp:46-66)
	virtual const char* getVariable( const std::string& name_) const
	{
		uint32_t id = sp_matcher_variable_id( m_h, name_.c_str());
		return id ? sp_matcher_variable_name( m_h, id) : 0;
	}
	virtual void defineTermFrequency( unsigned int termid, double df)
	{ check( sp_matcher_define_term_frequency( m_h, termid, df), "failed to define term frequency: %s"); }
	virtual void pushTerm( unsigned int termid)
	{ check( sp_matcher_push_term( m_h, termid), "failed to push term on the pattern match expression stack: %s"); }
	virtual void pushExpression( JoinOperation operation, std::size_t argc, unsigned int range, unsigned int cardinality)
	{
		static const int opmap[] = {SP_OP_SEQUENCE, SP_OP_SEQUENCE_IMM, SP_OP_SEQUENCE_STRUCT, SP_OP_WITHIN, SP_OP_WITHIN_STRUCT, SP_OP_ANY, SP_OP_AND};
		check( sp_matcher_push_expression( m_h, opmap[ (int)operation], argc, range, cardinality), "failed to push expression on the pattern match expression stack: %s");
	}
	virtual void pushPattern( const std::string& name_)
	{ check( sp_matcher_push_pattern( m_h, name_.c_str()), "failed to push pattern reference on the pattern match expression stack: %s"); }
	virtual void attachVariable( const std::string& name_)
	{ check( sp_matcher_attach_variable( m_h, name_.c_str()), "failed to attach variable to top element of the pattern match expression stack: %s"); }
	virtual void definePattern( const std::string& name_, const std::string& formatstring, bool visible)
	{
		// src/patternMatcher.cpp:561-566: format handle = position in the list of format strings, as in the C-ABI
		const PatternResultFormat* fmt = 0;
		if (!formatstring.empty())
		{
			try { fmt = m_resultFormatTable.createResultFormat( formatstring.c_str()); }
			catch (const std::bad_alloc&) { m_errorhnd->report( ErrorCodeOutOfMem, "memory allocation error in %s", "strus pattern"); return; }
			if (!fmt) return;
		}
		// A failing definePattern may or may not have taken a format handle (the reference pushes it before the check of
		// src/patternMatcher.cpp:576-579 but after the one of :549-552): the compiler behind the C-ABI says which, so
		// that this list and its list of format strings stay in step whatever the outcome.
		const uint32_t before = sp_matcher_format_count( m_h);
		check( sp_matcher_define_pattern( m_h, name_.c_str(), formatstring.c_str(), visible ? 1 : 0), "failed to close pattern definition on the pattern match expression stack: %s");
		if (sp_matcher_format_count( m_h) > before) m_resultFormatHandles.push_back( fmt);
	}
	virtual PatternMatcherContextInterface* createContext() const
	{
		sp_matcher_ctx_t* c = sp_matcher_ctx_create( m_h, nextDevice());
		if (!c) { m_errorhnd->report( ErrorCodeRuntimeError, "failed to create pattern match context: %s", sp_matcher_last_error( m_h)); return 0; }
		try { return new MatcherContext( c, m_h, &m_resultFormatHandles, m_errorhnd); }
		catch (const std::bad_alloc&) { sp_matcher_ctx_free( c); m_errorhnd->report( ErrorCodeOutOfMem, "memory allocation error in %s", "strus pattern"); return 0; }
	}
	virtual void defineOption( const std::string& name_, double value)
	{ check( sp_matcher_define_option( m_h, name_.c_str(), value), "failed to define pattern matching automaton option: %s"); }
	virtual bool compile()
	{ return check( sp_matcher_compile( m_h), "failed to compile (optimize) pattern matching automaton: %s"); }
	virtual const char* name() const { return "std"; }
	virtual StructView view() const { return StructView()( "name", name()); }
private:
	bool check( int rc, const char* fmt) const
	{
		if (rc == SP_OK) return true;
		m_errorhnd->report( codeOf( rc), fmt, sp_matcher_last_error( m_h));
		return false;
	}
	sp_matcher_t* m_h;
	ErrorBufferInterface* m_errorhnd;
	PatternResultFormatTable m_resultFormatTable;
	std::vector<const PatternResultFormat*> m_resultFormatHandles;
};

class Matcher :public PatternMatcherInterface
{
public:
	explicit Matcher( ErrorBufferInterface* e) :m_errorhnd(e){}
	virtual std::vector<std::string> getCompileOptionNames() const
	{
		static const char* ar[] = {"stopwordOccurrenceFactor", "weightFactor", "maxRange", "exclusive", "maxResultSize", 0};	// src/patternMatcher.cpp:707-716
		std::vector<std::string> rt;
		for (int i=0; ar[i]; ++i) rt.push_back( ar[i]);
		return rt;
	}
	virtual PatternMatcherInstanceInterface* createInstance() const
	{
		try { return new MatcherInstance( m_errorhnd); }
		catch (const std::bad_alloc&) { m_errorhnd->report( ErrorCodeOutOfMem, "failed to create pattern match instance: %s", "out of memory"); return 0; }
	}
	virtual const char* name() const { return "std"; }
	virtual StructView view() const { return StructView()( "name", name())( "description", "Token pattern matcher with a wavefront-per-document rule automaton on AMD CDNA4 GPUs"); }
private:
	ErrorBufferInterface* m_errorhnd;
};

} // namespace

#define DLL_PUBLIC __attribute__((visibility("default")))

DLL_PUBLIC PatternMatcherInterface* strus::createPatternMatcher_std( ErrorBufferInterface* errorhnd)
{
	try { return new Matcher( errorhnd); }
	catch (const std::bad_alloc&) { errorhnd->report( ErrorCodeOutOfMem, "error creating token pattern match interface: %s", "out of memory"); return 0; }
}

DLL_PUBLIC PatternLexerInterface* strus::createPatternLexer_std( ErrorBufferInterface* errorhnd)
{
	try { return new Lexer( errorhnd); }
	catch (const std::bad_alloc&) { errorhnd->report( ErrorCodeOutOfMem, "error creating char regex match interface: %s", "out of memory"); return 0; }
}
