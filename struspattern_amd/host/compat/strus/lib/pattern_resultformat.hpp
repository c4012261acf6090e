// Stand-in for strusAnalyzer's include/strus/lib/pattern_resultformat.hpp so that the host shim builds
// without strus installed.  The reference uses exactly these three classes (src/patternMatcher.cpp:22,
// :46, :79, :85-86, :115, :172-181, :253-262, :337, :561-566); in a real integration the shim is
// compiled against the real header and this file is not used.  The mini-language is the documented one
// (doc/webpage/introduction_struspattern.htm:143-161): "{variable}" / "{variable|separator}", anything
// else literal text.  An argument without a value of its own is a reference to the source text it
// covers; the real formatter encodes such references into the returned string for the analyzer to
// resolve (PatternResultFormatChunk), and so does this one -- the byte encoding of the real library is
// not known here, this one is "\x01<seg> <pos> <endseg> <endpos>\x01".
#ifndef STRUS_COMPAT_PATTERN_RESULTFORMAT_HPP
#define STRUS_COMPAT_PATTERN_RESULTFORMAT_HPP
#include "strus/errorBufferInterface.hpp"
#include "strus/analyzer/patternMatcherResult.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>

namespace strus {

class PatternResultFormatVariableMap
{
public:
	virtual ~PatternResultFormatVariableMap(){}
	virtual const char* getVariable( const std::string& name) const=0;	// canonical name or NULL if undefined
};

class PatternResultFormat
{
public:
	struct Element { bool isVariable; std::string value; std::string separator; };
	std::vector<Element> elements;
};

class PatternResultFormatTable
{
public:
	PatternResultFormatTable( const PatternResultFormatVariableMap* variableMap_, ErrorBufferInterface* errorhnd_)
		:m_variableMap(variableMap_),m_errorhnd(errorhnd_){}
	// returns NULL (error reported) on a syntax error or an unknown variable
	const PatternResultFormat* createResultFormat( const char* src)
	{
		PatternResultFormat fmt;
		std::string lit;
		for (char const* si=src; *si;)
		{
			if (*si == '\\' && si[1]) { lit.push_back( si[1]); si += 2; }
			else if (*si == '{')
			{
				const char* end = std::strchr( si, '}');
				if (!end) { m_errorhnd->report( ErrorCodeSyntax, "missing '}' in result format string '%s'", src); return 0; }
				if (!lit.empty()) { PatternResultFormat::Element e; e.isVariable = false; e.value = lit; fmt.elements.push_back( e); lit.clear(); }
				std::string body( si+1, end), var( body), sep( " ");
				std::size_t bar = body.find( '|');
				if (bar != std::string::npos) { var = body.substr( 0, bar); sep = body.substr( bar+1); }
				while (!var.empty() && (unsigned char)var[ var.size()-1] <= 32) var.resize( var.size()-1);
				while (!var.empty() && (unsigned char)var[ 0] <= 32) var.erase( 0, 1);
				const char* canonical = var.empty() ? 0 : m_variableMap->getVariable( var);
				if (!canonical) { m_errorhnd->report( ErrorCodeSyntax, "unknown variable '%s' in result format string", var.c_str()); return 0; }
				PatternResultFormat::Element e; e.isVariable = true; e.value = canonical; e.separator = sep;
				fmt.elements.push_back( e);
				si = end+1;
			}
			else lit.push_back( *si++);
		}
		if (!lit.empty()) { PatternResultFormat::Element e; e.isVariable = false; e.value = lit; fmt.elements.push_back( e); }
		m_formats.push_back( fmt);
		return &m_formats.back();
	}
private:
	const PatternResultFormatVariableMap* m_variableMap;
	ErrorBufferInterface* m_errorhnd;
	std::list<PatternResultFormat> m_formats;
};

class PatternResultFormatContext
{
public:
	explicit PatternResultFormatContext( ErrorBufferInterface* errorhnd_) :m_errorhnd(errorhnd_){}
	// the returned string lives as long as the context
	const char* map( const PatternResultFormat* fmt, const analyzer::PatternMatcherResultItem* ar, std::size_t arsize)
	{
		std::string out;
		for (std::size_t ei=0; ei<fmt->elements.size(); ++ei)
		{
			const PatternResultFormat::Element& e = fmt->elements[ ei];
			if (!e.isVariable) { out.append( e.value); continue; }
			int cnt = 0;
			for (std::size_t ai=0; ai<arsize; ++ai)
			{
				if (e.value != ar[ ai].name()) continue;
				if (cnt++) out.append( e.separator);
				if (ar[ ai].value()) out.append( ar[ ai].value());
				else
				{
					char buf[ 96];
					std::snprintf( buf, sizeof(buf), "\x01%d %d %d %d\x01", ar[ ai].origpos().seg(), ar[ ai].origpos().ofs(), ar[ ai].origend().seg(), ar[ ai].origend().ofs());
					out.append( buf);
				}
			}
		}
		m_strings.push_back( out);
		return m_strings.back().c_str();
	}
	void clear() { m_strings.clear(); }
private:
	ErrorBufferInterface* m_errorhnd;
	std::list<std::string> m_strings;
};

// iterator over a mapped value: literal chunks (value != 0) and source references (value == 0)
struct PatternResultFormatChunk
{
	const char* value; std::size_t valuesize;
	int start_seg, start_pos, end_seg, end_pos;

	static bool parseNext( PatternResultFormatChunk& result, char const*& src)
	{
		if (!*src) return false;
		if (*src == '\x01')
		{
			result.value = 0; result.valuesize = 0;
			if (4 != std::sscanf( src+1, "%d %d %d %d", &result.start_seg, &result.start_pos, &result.end_seg, &result.end_pos)) return false;
			const char* end = std::strchr( src+1, '\x01');
			if (!end) return false;
			src = end+1;
			return true;
		}
		result.value = src; result.start_seg = result.start_pos = result.end_seg = result.end_pos = 0;
		while (*src && *src != '\x01') ++src;
		result.valuesize = src - result.value;
		return true;
	}
};

} // namespace
#endif
