// Ports of the reference's two known-answer tests to the C++ plugin interface of this build:
// tests/simpleTokenPatternMatch/src/testSimpleTokenPatternMatch.cpp (6 sequence rules, 5 expected matches)
// tests/charRegexMatch/src/testCharRegexMatch.cpp case 1 (36 expected lexems).
// The expected values are read from the fixture files in tests/golden/ (argv[1], argv[2], flat
// text form written by tests/test_host_shim.py) so that no reference test source lives here.
// Also loads modstrus_analyzer_pattern.so through dlopen/dlsym("entryPoint") like the strus
// module loader does.
#include "strus/lib/pattern.hpp"
#include "strus/errorBufferInterface.hpp"
#include "strus/patternLexerInterface.hpp"
#include "strus/patternMatcherInterface.hpp"
#include "strus/analyzerModule.hpp"
#include "strus/lib/pattern_resultformat.hpp"
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <iostream>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

class ErrorBuffer :public strus::ErrorBufferInterface
{
public:
	ErrorBuffer() :m_has(false) { m_msg[0] = 0; }
	virtual void report( int, const char* format, ...)
	{
		if (m_has) return;
		va_list ap; va_start( ap, format); vsnprintf( m_msg, sizeof(m_msg), format, ap); va_end( ap);
		m_has = true;
	}
	virtual bool hasError() const { return m_has; }
	virtual const char* fetchError() { m_has = false; return m_msg; }
private:
	bool m_has; char m_msg[ 2048];
};

static std::vector<std::string> readLines( const char* path)
{
	std::ifstream in( path);
	if (!in) throw std::runtime_error( std::string("cannot open ") + path);
	std::vector<std::string> rt; std::string ln;
	while (std::getline( in, ln)) rt.push_back( ln);
	return rt;
}

// fixture line formats (tab separated):
//   simple:  RULE name term1 var1 term2 var2 range   |  DOC id ordpos origpos  |  EXPECT name ordpos
//   regex:   OPTION name | LEXEM id expr resultIndex level haspos | SYMBOL symid patid name | SRC text | EXPECT id ordpos origpos origsize
static std::vector<std::string> split( const std::string& s)
{
	std::vector<std::string> rt; std::string cur;
	for (size_t i=0; i<=s.size(); ++i) { if (i == s.size() || s[i] == '\t') { rt.push_back( cur); cur.clear(); } else cur.push_back( s[i]); }
	return rt;
}

static void testSimpleTokenPatternMatch( strus::PatternMatcherInterface* pt, ErrorBuffer& err, const char* fixture)
{
	std::unique_ptr<strus::PatternMatcherInstanceInterface> inst( pt->createInstance());
	if (!inst.get()) throw std::runtime_error( "failed to create pattern matcher instance");
	std::vector<std::string> lines = readLines( fixture);
	std::set<std::pair<std::string,unsigned> > expected;
	std::vector<std::vector<std::string> > doc;
	for (size_t i=0; i<lines.size(); ++i)
	{
		std::vector<std::string> f = split( lines[i]);
		if (f[0] == "RULE")
		{
			inst->pushTerm( atoi( f[2].c_str())); inst->attachVariable( f[3]);
			inst->pushTerm( atoi( f[4].c_str())); inst->attachVariable( f[5]);
			inst->pushExpression( strus::PatternMatcherInstanceInterface::OpSequence, 2, atoi( f[6].c_str()), 0);
			inst->definePattern( f[1], "", true);
		}
		else if (f[0] == "DOC") doc.push_back( f);
		else if (f[0] == "EXPECT") expected.insert( std::make_pair( f[1], (unsigned)atoi( f[2].c_str())));
	}
	inst->compile();
	if (err.hasError()) throw std::runtime_error( std::string("error creating automaton: ") + err.fetchError());
	std::unique_ptr<strus::PatternMatcherContextInterface> ctx( inst->createContext());
	if (!ctx.get()) throw std::runtime_error( std::string("failed to create context: ") + err.fetchError());
	for (size_t i=0; i<doc.size(); ++i)
	{
		ctx->putInput( strus::analyzer::PatternLexem( atoi( doc[i][1].c_str()), atoi( doc[i][2].c_str()), strus::analyzer::Position( 0, atoi( doc[i][3].c_str())), 1));
		if (err.hasError()) throw std::runtime_error( std::string("error matching rules: ") + err.fetchError());
	}
	std::vector<strus::analyzer::PatternMatcherResult> results = ctx->fetchResults();
	if (err.hasError()) throw std::runtime_error( std::string("error fetching results: ") + err.fetchError());
	std::set<std::pair<std::string,unsigned> > got;
	for (size_t i=0; i<results.size(); ++i) got.insert( std::make_pair( std::string( results[i].name()), results[i].ordpos()));
	if (got != expected) throw std::runtime_error( "simpleTokenPatternMatch: match set differs from the expected one");
	double installed = -1;
	strus::analyzer::PatternMatcherStatistics st = ctx->getStatistics();
	for (size_t i=0; i<st.items().size(); ++i) if (!strcmp( st.items()[i].name(), "nofProgramsInstalled")) installed = st.items()[i].value();
	std::cerr << "simpleTokenPatternMatch OK (" << results.size() << " results, " << installed << " programs installed)" << std::endl;
}

static void testCharRegexMatch( strus::PatternLexerInterface* pl, ErrorBuffer& err, const char* fixture)
{
	std::unique_ptr<strus::PatternLexerInstanceInterface> inst( pl->createInstance());
	if (!inst.get()) throw std::runtime_error( "failed to create lexer instance");
	std::vector<std::string> lines = readLines( fixture);
	std::string src;
	std::vector<std::vector<int> > expected;
	for (size_t i=0; i<lines.size(); ++i)
	{
		std::vector<std::string> f = split( lines[i]);
		if (f[0] == "OPTION") inst->defineOption( f[1], 0);
		else if (f[0] == "LEXEM") inst->defineLexem( atoi( f[1].c_str()), f[2], atoi( f[3].c_str()), atoi( f[4].c_str()), atoi( f[5].c_str()) ? strus::analyzer::BindContent : strus::analyzer::BindPredecessor);
		else if (f[0] == "SYMBOL") inst->defineSymbol( atoi( f[1].c_str()), atoi( f[2].c_str()), f[3]);
		else if (f[0] == "SRC") src = f[1];
		else if (f[0] == "EXPECT") { std::vector<int> e; for (int k=1; k<=4; ++k) e.push_back( atoi( f[k].c_str())); expected.push_back( e); }
	}
	if (!inst->compile()) throw std::runtime_error( std::string("error building term match automaton: ") + err.fetchError());
	std::unique_ptr<strus::PatternLexerContextInterface> ctx( inst->createContext());
	if (!ctx.get()) throw std::runtime_error( std::string("failed to create context: ") + err.fetchError());
	std::vector<strus::analyzer::PatternLexem> result = ctx->match( src.c_str(), src.size());
	if (result.empty() && err.hasError()) throw std::runtime_error( std::string("error matching: ") + err.fetchError());
	if (result.size() != expected.size()) throw std::runtime_error( "charRegexMatch: number of lexems differs");
	for (size_t i=0; i<result.size(); ++i)
	{
		if ((int)result[i].id() != expected[i][0] || (int)result[i].ordpos() != expected[i][1]
		||  result[i].origpos().ofs() != expected[i][2] || (int)result[i].origsize() != expected[i][3])
			throw std::runtime_error( "charRegexMatch: test failed");
	}
	// error convention: a broken expression is reported through the error buffer, compile() returns false
	std::unique_ptr<strus::PatternLexerInstanceInterface> bad( pl->createInstance());
	bad->defineLexem( 1, "(unbalanced", 0, 1, strus::analyzer::BindContent);
	if (bad->compile() || !err.hasError()) throw std::runtime_error( "charRegexMatch: compile error not reported");
	std::cerr << "charRegexMatch OK (" << result.size() << " lexems); compile error text: " << err.fetchError() << std::endl;
}

// Result format strings through the C++ interface (src/patternMatcher.cpp:164-190, :248-269): the money
// example of the reference's web page (doc/webpage/introduction_struspattern.htm:146-160) on token ids.
static std::string resolveValue( const char* value, const std::string& src)
{
	std::string rt;
	strus::PatternResultFormatChunk chunk;
	char const* vi = value;
	while (strus::PatternResultFormatChunk::parseNext( chunk, vi))
	{
		if (chunk.value) rt.append( chunk.value, chunk.valuesize);
		else rt.append( src, chunk.start_pos, chunk.end_pos - chunk.start_pos);
	}
	return rt;
}

static void testResultFormat( strus::PatternMatcherInterface* pt, ErrorBuffer& err)
{
	enum {AMOUNT=1, CURRENCY_CHF=2, CURRENCY_EUR=3};
	const std::string src = "3'645 Eur";
	std::unique_ptr<strus::PatternMatcherInstanceInterface> inst( pt->createInstance());
	inst->pushTerm( CURRENCY_CHF); inst->definePattern( "Currency", "CHF", true);
	inst->pushTerm( CURRENCY_EUR); inst->definePattern( "Currency", "EUR", true);
	inst->pushTerm( AMOUNT); inst->attachVariable( "value");
	inst->pushPattern( "Currency"); inst->attachVariable( "currency");
	inst->pushExpression( strus::PatternMatcherInstanceInterface::OpSequenceImm, 2, 2, 0);
	inst->definePattern( "MoneyAmount", "{currency} {value}", true);
	inst->pushTerm( AMOUNT); inst->attachVariable( "value");
	inst->pushPattern( "Currency"); inst->attachVariable( "currency");
	inst->pushExpression( strus::PatternMatcherInstanceInterface::OpSequenceImm, 2, 2, 0);
	inst->definePattern( "Plain", "", true);
	inst->compile();
	if (err.hasError()) throw std::runtime_error( std::string("resultFormat: error creating automaton: ") + err.fetchError());
	std::unique_ptr<strus::PatternMatcherContextInterface> ctx( inst->createContext());
	if (!ctx.get()) throw std::runtime_error( std::string("failed to create context: ") + err.fetchError());
	ctx->putInput( strus::analyzer::PatternLexem( AMOUNT, 1, strus::analyzer::Position( 0, 0), 5));
	ctx->putInput( strus::analyzer::PatternLexem( CURRENCY_EUR, 2, strus::analyzer::Position( 0, 6), 3));
	std::vector<strus::analyzer::PatternMatcherResult> results = ctx->fetchResults();
	if (err.hasError()) throw std::runtime_error( std::string("resultFormat: error fetching results: ") + err.fetchError());
	int seen = 0;
	for (size_t i=0; i<results.size(); ++i)
	{
		const strus::analyzer::PatternMatcherResult& r = results[i];
		std::string name( r.name());
		if (name == "Currency")
		{
			if (!r.value() || std::string( r.value()) != "EUR" || !r.items().empty()) throw std::runtime_error( "resultFormat: constant format string");
			seen |= 1;
		}
		else if (name == "MoneyAmount")
		{
			// the page: an input "3'645 Eur" would generate a value "EUR 3'645"
			if (!r.value() || resolveValue( r.value(), src) != "EUR 3'645" || !r.items().empty()) throw std::runtime_error( "resultFormat: variable substitution");
			seen |= 2;
		}
		else if (name == "Plain")
		{
			if (r.value() || r.items().size() != 2) throw std::runtime_error( "resultFormat: result without a format string");
			const strus::analyzer::PatternMatcherResultItem& cur = r.items()[0];
			const strus::analyzer::PatternMatcherResultItem& val = r.items()[1];
			if (std::string( cur.name()) != "currency" || !cur.value() || std::string( cur.value()) != "EUR") throw std::runtime_error( "resultFormat: item value of a formatted sub-pattern");
			if (std::string( val.name()) != "value" || val.value() || val.origpos().ofs() != 0 || val.origend().ofs() != 5) throw std::runtime_error( "resultFormat: plain item");
			seen |= 4;
		}
	}
	if (seen != 7 || results.size() != 3) throw std::runtime_error( "resultFormat: unexpected result set");
	// an unknown variable in a format string is an error of definePattern
	std::unique_ptr<strus::PatternMatcherInstanceInterface> bad( pt->createInstance());
	bad->pushTerm( 1); bad->definePattern( "X", "{nosuchvariable}", true);
	if (!err.hasError()) throw std::runtime_error( "resultFormat: unknown variable not reported");
	(void)err.fetchError();
	std::cerr << "resultFormat OK (" << results.size() << " results)" << std::endl;
}

// ---- the plugin path under the reference's threading model: one Context per thread over a shared Instance
// (tests/randomTokenPatternMatch/src/testRandomTokenPatternMatch.cpp:325-345): N threads, each with its own lexer context and
// matcher context, every thread runs every document (match -> putInput per lexem -> fetchResults) `repeat` times.
// Fixture (tab separated, written by tests/test_host_shim.py): OPTION name | LEXEM id expr resultIndex level haspos |
// XRULE name op range delim nterms t1 v1 .. | DOC hex(text).  Output: one line per document "DOCSUM index nresults fnv1a64" (what thread 0
// got; the other threads must get the same) and "THREADS n docs d bytes b seconds s".
static uint64_t fnv( uint64_t h, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i=0; i<n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }
static uint64_t fnvU( uint64_t h, uint64_t v) { return fnv( h, &v, sizeof(v)); }
static uint64_t fnvS( uint64_t h, const char* s) { return fnv( fnv( h, s, strlen( s)), "\0", 1); }

static int runThreads( int nthreads, const char* fixture, int repeat)
{
	ErrorBuffer err;
	std::unique_ptr<strus::PatternMatcherInterface> pt( strus::createPatternMatcher_std( &err));
	std::unique_ptr<strus::PatternLexerInterface> pl( strus::createPatternLexer_std( &err));
	std::unique_ptr<strus::PatternLexerInstanceInterface> li( pl->createInstance());
	std::unique_ptr<strus::PatternMatcherInstanceInterface> mi( pt->createInstance());
	std::vector<std::string> docs;
	std::vector<std::string> lines = readLines( fixture);
	for (size_t i=0; i<lines.size(); ++i)
	{
		std::vector<std::string> f = split( lines[i]);
		if (f[0] == "OPTION") li->defineOption( f[1], 0);
		else if (f[0] == "LEXEM") li->defineLexem( atoi( f[1].c_str()), f[2], atoi( f[3].c_str()), atoi( f[4].c_str()), atoi( f[5].c_str()) ? strus::analyzer::BindContent : strus::analyzer::BindPredecessor);
		else if (f[0] == "XRULE")
		{
			strus::PatternMatcherInstanceInterface::JoinOperation op = strus::PatternMatcherInstanceInterface::OpSequence;
			if (f[2] == "within") op = strus::PatternMatcherInstanceInterface::OpWithin;
			else if (f[2] == "sequence_struct") op = strus::PatternMatcherInstanceInterface::OpSequenceStruct;
			else if (f[2] == "within_struct") op = strus::PatternMatcherInstanceInterface::OpWithinStruct;
			else if (f[2] == "any") op = strus::PatternMatcherInstanceInterface::OpAny;
			const unsigned delim = (unsigned)strtoul( f[4].c_str(), 0, 10);
			const int nterms = atoi( f[5].c_str());
			int argc = nterms;
			if (delim) { mi->pushTerm( delim); ++argc; }
			for (int t=0; t<nterms; ++t) { mi->pushTerm( (unsigned)strtoul( f[6+2*t].c_str(), 0, 10)); mi->attachVariable( f[7+2*t]); }
			mi->pushExpression( op, argc, atoi( f[3].c_str()), 0);
			mi->definePattern( f[1], "", true);
		}
		else if (f[0] == "DOC")
		{
			std::string text;
			for (size_t k=0; k+1<f[1].size(); k+=2) text.push_back( (char)strtoul( f[1].substr( k, 2).c_str(), 0, 16));
			docs.push_back( text);
		}
	}
	if (!li->compile() || !mi->compile() || err.hasError()) throw std::runtime_error( std::string("threads: compile failed: ") + err.fetchError());
	std::vector<std::vector<uint64_t> > sums( nthreads, std::vector<uint64_t>( 2*docs.size(), 0));
	std::vector<std::string> failures( nthreads);
	std::atomic<int> ready( 0);
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	auto work = [&]( int ti)
	{
		try
		{
			ErrorBuffer terr;		// (one error buffer slot per thread in the reference; a private buffer here)
			std::unique_ptr<strus::PatternLexerContextInterface> lc( li->createContext());
			std::unique_ptr<strus::PatternMatcherContextInterface> mc( mi->createContext());
			if (!lc.get() || !mc.get()) throw std::runtime_error( "createContext failed");
			for (int rep=-1; rep<repeat; ++rep)
			{
				if (rep == 0)
				{
					// the first pass (rep -1) sized every context's buffers; the timed passes start together
					ready.fetch_add( 1);
					while (ready.load() < nthreads) std::this_thread::yield();
					if (ti == 0) t0 = std::chrono::steady_clock::now();
				}
				for (size_t di=0; di<docs.size(); ++di)
				{
					std::vector<strus::analyzer::PatternLexem> lex = lc->match( docs[ di].c_str(), docs[ di].size());
					mc->reset();
					for (size_t k=0; k<lex.size(); ++k) mc->putInput( lex[ k]);
					std::vector<strus::analyzer::PatternMatcherResult> res = mc->fetchResults();
					uint64_t h = 1469598103934665603ull;
					for (size_t r=0; r<res.size(); ++r)
					{
						h = fnvS( h, res[ r].name()); h = fnvU( h, res[ r].ordpos()); h = fnvU( h, res[ r].ordend());
						h = fnvU( h, res[ r].origpos().ofs()); h = fnvU( h, res[ r].origend().ofs());
						for (size_t q=0; q<res[ r].items().size(); ++q)
						{
							const strus::analyzer::PatternMatcherResultItem& it = res[ r].items()[ q];
							h = fnvS( h, it.name()); h = fnvU( h, it.ordpos()); h = fnvU( h, it.ordend()); h = fnvU( h, it.origpos().ofs()); h = fnvU( h, it.origend().ofs());
						}
					}
					sums[ ti][ 2*di] = res.size(); sums[ ti][ 2*di+1] = h;
				}
			}
		}
		catch (const std::exception& e) { failures[ ti] = e.what(); ready.fetch_add( 1); /*(the others must not wait for a thread that has given up)*/ }
	};
	std::vector<std::thread> th;
	for (int ti=0; ti<nthreads; ++ti) th.push_back( std::thread( work, ti));
	for (int ti=0; ti<nthreads; ++ti) th[ ti].join();
	const double secs = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0).count();
	if (err.hasError()) throw std::runtime_error( std::string("threads: ") + err.fetchError());
	for (int ti=0; ti<nthreads; ++ti)
	{
		if (!failures[ ti].empty()) throw std::runtime_error( "threads: thread failed: " + failures[ ti]);
		if (sums[ ti] != sums[ 0]) throw std::runtime_error( "threads: the threads disagree");
	}
	size_t bytes = 0;
	for (size_t di=0; di<docs.size(); ++di) bytes += docs[ di].size();
	for (size_t di=0; di<docs.size(); ++di) std::cout << "DOCSUM\t" << di << "\t" << sums[ 0][ 2*di] << "\t" << sums[ 0][ 2*di+1] << std::endl;
	std::cout << "THREADS\t" << nthreads << "\tdocs\t" << (docs.size()*(size_t)repeat*(size_t)nthreads) << "\tbytes\t" << (bytes*(size_t)repeat*(size_t)nthreads) << "\tseconds\t" << secs << std::endl;
	return 0;
}

int main( int argc, const char** argv)
{
	try
	{
		ErrorBuffer err;
		if (argc == 2 && !strcmp( argv[1], "--options"))
		{
			// the option name lists of both interfaces, one per line (no GPU needed): tests/test_host_shim.py compares them
			// with the reference's (src/patternLexer.cpp:1154-1163, src/patternMatcher.cpp:707-716)
			std::unique_ptr<strus::PatternMatcherInterface> pt( strus::createPatternMatcher_std( &err));
			std::unique_ptr<strus::PatternLexerInterface> pl( strus::createPatternLexer_std( &err));
			if (!pt.get() || !pl.get()) throw std::runtime_error( "failed to create the interfaces");
			std::vector<std::string> lo = pl->getCompileOptionNames(), mo = pt->getCompileOptionNames();
			for (size_t i=0; i<lo.size(); ++i) std::cout << "lexer\t" << lo[i] << std::endl;
			for (size_t i=0; i<mo.size(); ++i) std::cout << "matcher\t" << mo[i] << std::endl;
			return 0;
		}
		if (argc == 5 && !strcmp( argv[1], "--threads")) return runThreads( atoi( argv[2]), argv[3], atoi( argv[4]));
		if (argc < 4) { std::cerr << "usage: testStrusInterface <simple fixture> <regex fixture> <module .so> | --options | --threads <n> <fixture> <repeat>" << std::endl; return 2; }
		// 1. through the exported factory functions (libstrus_pattern)
		std::unique_ptr<strus::PatternMatcherInterface> pt( strus::createPatternMatcher_std( &err));
		std::unique_ptr<strus::PatternLexerInterface> pl( strus::createPatternLexer_std( &err));
		if (!pt.get() || !pl.get()) throw std::runtime_error( "failed to create the interfaces");
		testSimpleTokenPatternMatch( pt.get(), err, argv[1]);
		testCharRegexMatch( pl.get(), err, argv[2]);
		testResultFormat( pt.get(), err);
		// 2. through the module entry point, like the strus module loader
		void* mod = dlopen( argv[3], RTLD_NOW | RTLD_LOCAL);
		if (!mod) throw std::runtime_error( std::string("dlopen failed: ") + dlerror());
		strus::AnalyzerModule* ep = (strus::AnalyzerModule*)dlsym( mod, "entryPoint");
		if (!ep) throw std::runtime_error( "module has no symbol entryPoint");
		if (strcmp( ep->patternLexer->name, "std") || strcmp( ep->patternMatcher->name, "std")) throw std::runtime_error( "module constructors are not named std");
		std::unique_ptr<strus::PatternMatcherInterface> pt2( ep->patternMatcher->create( &err));
		std::unique_ptr<strus::PatternLexerInterface> pl2( ep->patternLexer->create( &err));
		testSimpleTokenPatternMatch( pt2.get(), err, argv[1]);
		testCharRegexMatch( pl2.get(), err, argv[2]);
		std::cerr << "module entryPoint OK: " << ep->version3rdparty << std::endl;
		std::cout << "OK" << std::endl;
		return 0;
	}
	catch (const std::exception& e)
	{
		std::cerr << "ERROR " << e.what() << std::endl;
		return 1;
	}
}
