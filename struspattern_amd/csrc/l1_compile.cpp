// Level-1 lexer compiler, see l1_compile.hpp and l1_tables.h.
//
// Pipeline per expression:  text --(Syntax)--> tree of byte sets / operators / assertions
//   --(Glushkov with conditional edges)--> positions, first/last/follow sets whose members carry the
//   set of zero-width assertions crossed --(context splitting)--> unconditional follow edges,
//   context-dependent start and accept sets --(layout)--> bit ranges in 64-bit words.
// Reported semantics (what the reference obtains from Hyperscan with HS_FLAG_SOM_LEFTMOST|HS_FLAG_UTF8,
// src/patternLexer.cpp:391-405): every end offset of a non-empty match once, with its leftmost start.
#include "l1_compile.hpp"
#include "serial.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <set>
#include <stdexcept>

using namespace spa;

namespace {

typedef std::vector<std::pair<uint32_t,uint32_t> > CpRanges;

#include "unicode_categories.inc"


// A leaf of the expression tree: a set of byte values.  A leaf with cpRef >= 0 stands for the lead byte of a
// character whose code point is in the registered set cpRef (all of one UTF-8 length); its byte set then holds
// the lead bytes such characters begin with, and positions of that kind get their class from the code point
// (LexTables::cpBlocks / cpPages) instead of from the byte.
struct ByteSet
{
	uint64_t w[4];
	int cpRef;
	ByteSet() :cpRef(-1) { w[0]=w[1]=w[2]=w[3]=0; }
	void add( unsigned c) { w[ c>>6] |= (1ull << (c&63)); }
	void addRange( unsigned lo, unsigned hi) { for (unsigned c=lo; c<=hi; ++c) add( c); }
	bool has( unsigned c) const { return (w[ c>>6] >> (c&63)) & 1ull; }
	bool empty() const { return !(w[0]|w[1]|w[2]|w[3]); }
};

// assertion kinds, combined as a bit set = conjunction
enum {A_WB=1, A_NWB=2, A_BOL=4, A_EOL=8, A_BOD=16, A_EOD=32,
      A_SAMEWORD=64};	// between the bytes of one multi-byte character (UCP): both sides are of a word character or both are not

enum TreeOp {T_EMPTY, T_SET, T_CAT, T_ALT, T_STAR, T_PLUS, T_OPT, T_ASSERT, T_GROUP};
struct Tree
{
	TreeOp op;
	ByteSet set;
	unsigned assertion;
	unsigned group;
	std::vector<Tree> kids;
	Tree() :op(T_EMPTY),assertion(0),group(0){}
	static Tree leaf( const ByteSet& s) { Tree t; t.op = T_SET; t.set = s; return t; }
	static Tree node( TreeOp o) { Tree t; t.op = o; return t; }
	static Tree cat( const std::vector<Tree>& k) { if (k.empty()) return Tree(); if (k.size() == 1) return k[0]; Tree t; t.op = T_CAT; t.kids = k; return t; }
	static Tree alt( const std::vector<Tree>& k) { if (k.size() == 1) return k[0]; Tree t; t.op = T_ALT; t.kids = k; return t; }
};

bool isWordChar( unsigned c) { return (c>='0'&&c<='9')||(c>='A'&&c<='Z')||(c>='a'&&c<='z')||c=='_'; }

// ---- UTF-8: code point ranges -> alternatives of byte-range sequences
int encodeUtf8( uint32_t cp, unsigned char* b)
{
	if (cp < 0x80) { b[0] = (unsigned char)cp; return 1; }
	if (cp < 0x800) { b[0] = (unsigned char)(0xC0 | (cp >> 6)); b[1] = (unsigned char)(0x80 | (cp & 0x3F)); return 2; }
	if (cp < 0x10000) { b[0] = (unsigned char)(0xE0 | (cp >> 12)); b[1] = (unsigned char)(0x80 | ((cp >> 6) & 0x3F)); b[2] = (unsigned char)(0x80 | (cp & 0x3F)); return 3; }
	b[0] = (unsigned char)(0xF0 | (cp >> 18)); b[1] = (unsigned char)(0x80 | ((cp >> 12) & 0x3F));
	b[2] = (unsigned char)(0x80 | ((cp >> 6) & 0x3F)); b[3] = (unsigned char)(0x80 | (cp & 0x3F)); return 4;
}
// `glue`: node put between the bytes of one character (empty tree = none)
void splitUtf8( uint32_t lo, uint32_t hi, std::vector<Tree>& alts, const Tree& glue)
{
	// cut at encoded-length boundaries, then at continuation-byte boundaries until every byte of the
	// sequence ranges independently
	static const uint32_t lenMax[3] = {0x7F, 0x7FF, 0xFFFF};
	for (int i=0; i<3; ++i)
	{
		if (lo <= lenMax[i] && hi > lenMax[i]) { splitUtf8( lo, lenMax[i], alts, glue); splitUtf8( lenMax[i]+1, hi, alts, glue); return; }
	}
	if (hi >= 0x80)
	{
		for (int i=1; i<=3; ++i)
		{
			uint32_t m = (1u << (6*i)) - 1;
			if ((lo & ~m) != (hi & ~m))
			{
				if (lo & m) { splitUtf8( lo, lo|m, alts, glue); splitUtf8( (lo|m)+1, hi, alts, glue); return; }
				if ((hi & m) != m) { splitUtf8( lo, (hi & ~m)-1, alts, glue); splitUtf8( hi & ~m, hi, alts, glue); return; }
			}
		}
	}
	unsigned char a[4], b[4];
	int n = encodeUtf8( lo, a); encodeUtf8( hi, b);
	std::vector<Tree> seq;
	for (int i=0; i<n; ++i) { ByteSet s; s.addRange( a[i], b[i]); if (i && glue.op != T_EMPTY) seq.push_back( glue); seq.push_back( Tree::leaf( s)); }
	alts.push_back( Tree::cat( seq));
}

// ---- syntax: expression text -> Tree
class Syntax
{
public:
	// sameWord: put A_SAMEWORD between the bytes of every multi-byte character (UCP expressions with \\b / \\B)
	Syntax( const std::string& text, unsigned options, std::vector<CpRanges>* cpSets, bool sameWord)
		:m_text(text),m_at(0),m_caseless((options & LEX_CASELESS)!=0),m_dotall((options & LEX_DOTALL)!=0)
		,m_multiline((options & LEX_MULTILINE)!=0),m_utf8(true),m_ucp((options & LEX_UCP)!=0),m_sameWord(sameWord),m_groups(0),m_cpSets(cpSets){}

	Tree run()
	{
		Tree t = alternation();
		if (m_at != m_text.size()) error( "unmatched ')'");
		return t;
	}
	unsigned groups() const {return m_groups;}

private:
	const std::string& m_text;
	size_t m_at;
	bool m_caseless, m_dotall, m_multiline, m_utf8, m_ucp, m_sameWord;
	unsigned m_groups;
	std::vector<CpRanges>* m_cpSets;

	void error( const std::string& what) const { throw std::runtime_error( "failed to compile pattern \"" + m_text + "\": " + what); }
	bool done() const { return m_at >= m_text.size(); }
	unsigned char cur() const { return (unsigned char)m_text[ m_at]; }
	bool lookingAt( char c) const { return !done() && m_text[ m_at] == c; }

	Tree alternation()
	{
		std::vector<Tree> branches;
		branches.push_back( sequence());
		while (lookingAt( '|')) { ++m_at; branches.push_back( sequence()); }
		return Tree::alt( branches);
	}
	Tree sequence()
	{
		std::vector<Tree> items;
		while (!done() && cur() != '|' && cur() != ')') items.push_back( quantified());
		return Tree::cat( items);
	}
	static Tree repeat( const Tree& x, int lo, int hi)
	{
		// x{lo,hi}: lo copies then nested optionals; x{lo,}: lo-1 copies then x+ (x* if lo == 0)
		std::vector<Tree> seq;
		if (hi < 0)
		{
			for (int i=0; i+1<lo; ++i) seq.push_back( x);
			Tree r = Tree::node( lo > 0 ? T_PLUS : T_STAR); r.kids.push_back( x);
			seq.push_back( r);
			return Tree::cat( seq);
		}
		for (int i=0; i<lo; ++i) seq.push_back( x);
		Tree tail;
		bool haveTail = false;
		for (int i=lo; i<hi; ++i)
		{
			std::vector<Tree> inner; inner.push_back( x);
			if (haveTail) inner.push_back( tail);
			Tree o = Tree::node( T_OPT); o.kids.push_back( Tree::cat( inner));
			tail = o; haveTail = true;
		}
		if (haveTail) seq.push_back( tail);
		return Tree::cat( seq);
	}
	bool bounds( int& lo, int& hi)
	{
		size_t p = m_at+1;
		auto number = [&]( int& v) -> bool {
			if (p >= m_text.size() || m_text[p] < '0' || m_text[p] > '9') return false;
			v = 0;
			while (p < m_text.size() && m_text[p] >= '0' && m_text[p] <= '9') { v = v*10 + (m_text[p]-'0'); if (v > 1000) error( "repeat count too large"); ++p; }
			return true;
		};
		if (!number( lo)) return false;
		hi = lo;
		if (p < m_text.size() && m_text[p] == ',')
		{
			++p;
			if (p < m_text.size() && m_text[p] == '}') hi = -1;
			else if (!number( hi)) return false;
		}
		if (p >= m_text.size() || m_text[p] != '}') return false;
		if (hi >= 0 && hi < lo) error( "repeat bounds out of order");
		m_at = p+1;
		return true;
	}
	Tree quantified()
	{
		Tree x = atom();
		while (!done())
		{
			int lo, hi;
			if (cur() == '*') { lo = 0; hi = -1; ++m_at; }
			else if (cur() == '+') { lo = 1; hi = -1; ++m_at; }
			else if (cur() == '?') { lo = 0; hi = 1; ++m_at; }
			else if (cur() == '{' && bounds( lo, hi)) {}
			else break;
			if (!done() && (cur() == '?' || cur() == '+')) ++m_at;	// lazy/possessive do not change the set of (from,to)
			if (x.op == T_ASSERT) error( "nothing to repeat");
			x = repeat( x, lo, hi);
		}
		return x;
	}

	// -- character sets
	uint32_t topCp() const { return m_utf8 ? 0x10FFFFu : 0xFFu; }
	static void norm( CpRanges& r)
	{
		std::sort( r.begin(), r.end());
		CpRanges o;
		for (size_t i=0; i<r.size(); ++i)
		{
			if (!o.empty() && r[i].first <= o.back().second + 1) o.back().second = std::max( o.back().second, r[i].second);
			else o.push_back( r[i]);
		}
		r.swap( o);
	}
	CpRanges complement( CpRanges r) const
	{
		norm( r);
		CpRanges o; uint32_t at = 0;
		for (size_t i=0; i<r.size(); ++i) { if (r[i].first > at) o.push_back( std::make_pair( at, r[i].first-1)); at = r[i].second+1; }
		if (at <= topCp()) o.push_back( std::make_pair( at, topCp()));
		return o;
	}
	void fold( CpRanges& r) const
	{
		// CASELESS: every code point of the set brings the other members of its case class (Unicode case folding,
		// csrc/unicode_categories.inc; in UTF-8 mode also beyond ASCII: K k KELVIN SIGN, S s LONG S, ...)
		if (!m_caseless) return;
		size_t n = r.size();
		for (size_t i=0; i<n; ++i)
		{
			const uint32_t lo = r[i].first, hi = r[i].second;
			size_t a = 0, b = UC_CASEPAIR_COUNT;
			while (a < b) { size_t mid = (a+b)/2; if (UC_CASEPAIRS[ mid].cp < lo) a = mid+1; else b = mid; }
			for (; a < UC_CASEPAIR_COUNT && UC_CASEPAIRS[ a].cp <= hi; ++a)
			{
				const uint32_t o = UC_CASEPAIRS[ a].other;
				if (o > topCp()) continue;
				if (!m_utf8 && UC_CASEPAIRS[ a].cp > 0x7F) continue;
				r.push_back( std::make_pair( o, o));
			}
		}
	}
	Tree glue() const { return m_sameWord ? assertNode( A_SAMEWORD) : Tree(); }
	// a then b as bytes of one character
	Tree inChar( const Tree& a, const Tree& b) const
	{
		std::vector<Tree> sq; sq.push_back( a); if (m_sameWord) sq.push_back( assertNode( A_SAMEWORD)); sq.push_back( b);
		return Tree::cat( sq);
	}
	Tree fromRanges( CpRanges r) const
	{
		norm( r);
		if (r.empty()) { ByteSet none; return Tree::leaf( none); }
		if (!m_utf8 || r.back().second < 0x80)
		{
			ByteSet s;
			for (size_t i=0; i<r.size(); ++i) s.addRange( r[i].first, std::min<uint32_t>( r[i].second, 255));
			return Tree::leaf( s);
		}
		std::vector<Tree> alts;
		CpRanges rest;			// what is neither ASCII nor "everything beyond ASCII"
		// merge all one-byte alternatives into a single leaf
		ByteSet ascii; bool haveAscii = false;
		for (size_t i=0; i<r.size(); ++i)
		{
			uint32_t lo = r[i].first, hi = r[i].second;
			if (lo < 0x80) { ascii.addRange( lo, std::min<uint32_t>( hi, 0x7F)); haveAscii = true; lo = 0x80; }
			if (lo > hi) continue;
			if (lo == 0x80 && hi >= 0x10FFFF)
			{
				// every character beyond ASCII (negated ASCII classes, '.'): on valid UTF-8 -- which the
				// reference's Hyperscan UTF-8 mode requires of its input too -- the lead byte alone
				// decides the length, so ((L4 C | L3) C | L2) C with shared continuation positions
				// (6 positions) reports exactly what the 26 positions of the exact range split report
				ByteSet l2, l3, l4, c;
				l2.addRange( 0xC2, 0xDF); l3.addRange( 0xE0, 0xEF); l4.addRange( 0xF0, 0xF4); c.addRange( 0x80, 0xBF);
				std::vector<Tree> a3; a3.push_back( inChar( Tree::leaf( l4), Tree::leaf( c))); a3.push_back( Tree::leaf( l3));
				std::vector<Tree> a2; a2.push_back( inChar( Tree::alt( a3), Tree::leaf( c))); a2.push_back( Tree::leaf( l2));
				alts.push_back( inChar( Tree::alt( a2), Tree::leaf( c)));
				continue;
			}
			rest.push_back( std::make_pair( lo, hi));
		}
		if (rest.size() <= 4)
		{
			for (size_t i=0; i<rest.size(); ++i) splitUtf8( rest[i].first, rest[i].second, alts, glue());
		}
		else
		{
			// a large set (a Unicode category): ((P4 C | P3) C | P2) C where Pn is the lead byte of an n-byte
			// character of the set -- told by its code point, see ByteSet::cpRef -- and C any continuation byte
			static const uint32_t lenLo[5] = {0,0,0x80,0x800,0x10000}, lenHi[5] = {0,0,0x7FF,0xFFFF,0x10FFFF};
			ByteSet c; c.addRange( 0x80, 0xBF);
			Tree level; bool have = false;
			for (int n=4; n>=2; --n)
			{
				CpRanges part;
				ByteSet leads;
				for (size_t i=0; i<rest.size(); ++i)
				{
					uint32_t lo = std::max( rest[i].first, lenLo[n]), hi = std::min( rest[i].second, lenHi[n]);
					if (lo > hi) continue;
					part.push_back( std::make_pair( lo, hi));		// (a negated set holds the surrogate range like any other; ED A0..BF xx decodes to it)
				}
				for (size_t i=0; i<part.size(); ++i)
				{
					unsigned char a[4], b[4]; encodeUtf8( part[i].first, a); encodeUtf8( part[i].second, b);
					leads.addRange( a[0], b[0]);
				}
				std::vector<Tree> branches;
				if (have) branches.push_back( inChar( level, Tree::leaf( c)));
				if (!part.empty())
				{
					leads.cpRef = (int)m_cpSets->size(); m_cpSets->push_back( part);
					branches.push_back( Tree::leaf( leads));
				}
				if (!branches.empty()) { level = Tree::alt( branches); have = true; }
			}
			if (have) alts.push_back( inChar( level, Tree::leaf( c)));
		}
		if (haveAscii) alts.insert( alts.begin(), Tree::leaf( ascii));
		return Tree::alt( alts);
	}
	// \\p{Name}, \\pL, \\P{..}, \\p{^..}: Unicode general category (m_at is behind the p / P)
	void property( bool negated, CpRanges& r)
	{
		std::string name;
		if (lookingAt( '{'))
		{
			size_t e = m_text.find( '}', m_at);
			if (e == std::string::npos) error( "unterminated \\p{..}");
			name = m_text.substr( m_at+1, e-m_at-1); m_at = e+1;
		}
		else if (!done()) { name = std::string( 1, (char)cur()); ++m_at; }
		if (!name.empty() && name[0] == '^') { negated = !negated; name.erase( 0, 1); }
		for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
		{
			if (name != c->name) continue;
			CpRanges t;
			for (uint32_t i=0; i<c->count; ++i) t.push_back( std::make_pair( c->ranges[i].lo, c->ranges[i].hi));
			if (negated) t = complement( t);
			r.insert( r.end(), t.begin(), t.end());
			return;
		}
		error( "unknown unicode property \\p{" + name + "}");
	}
	static void category( const char* name, CpRanges& r)
	{
		for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
		{
			if (std::strcmp( name, c->name)) continue;
			for (uint32_t i=0; i<c->count; ++i) r.push_back( std::make_pair( c->ranges[i].lo, c->ranges[i].hi));
		}
	}
	// \\d \\w \\s; with UCP by Unicode properties (PCRE: \\d = Nd, \\w = L | N | _, \\s = Z | \\h | \\v)
	void shorthand( char e, CpRanges& r) const
	{
		switch (e | 32)
		{
			case 'd': r.push_back( std::make_pair( '0','9')); if (m_ucp) category( "Nd", r); break;
			case 'w': r.push_back( std::make_pair( '0','9')); r.push_back( std::make_pair( 'A','Z')); r.push_back( std::make_pair( 'a','z')); r.push_back( std::make_pair( '_','_'));
				if (m_ucp) { category( "L", r); category( "N", r); }
				break;
			case 's': r.push_back( std::make_pair( 9, 13)); r.push_back( std::make_pair( ' ',' '));
				if (m_ucp) { category( "Z", r); r.push_back( std::make_pair( 0x85u, 0x85u)); }
				break;
		}
	}
	uint32_t literalChar()
	{
		unsigned char c = cur();
		if (!m_utf8 || c < 0x80) { ++m_at; return c; }
		int n = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 0;
		if (!n || m_at + n > m_text.size()) error( "invalid UTF-8 sequence");
		uint32_t cp = c & (0xFFu >> (n+1));
		for (int i=1; i<n; ++i) cp = (cp << 6) | ((unsigned char)m_text[ m_at+i] & 0x3F);
		m_at += n;
		return cp;
	}
	uint32_t escapedChar( unsigned char e)
	{
		switch (e)
		{
			case 'n': return 10; case 'r': return 13; case 't': return 9; case 'f': return 12; case 'v': return 11;
			case 'a': return 7; case 'e': return 27; case '0': return 0;
			case 'x':
			{
				uint32_t v = 0;
				for (int i=0; i<2; ++i)
				{
					if (done()) error( "incomplete \\x escape");
					unsigned char h = cur(); ++m_at;
					if (h >= '0' && h <= '9') v = v*16 + (h-'0');
					else if ((h|32) >= 'a' && (h|32) <= 'f') v = v*16 + ((h|32)-'a'+10);
					else error( "incomplete \\x escape");
				}
				return v;
			}
		}
		if ((e >= '0' && e <= '9') || (e >= 'A' && e <= 'Z') || (e >= 'a' && e <= 'z')) error( std::string("unsupported escape \\") + (char)e);
		return e;
	}
	Tree assertNode( unsigned kind) const { Tree t = Tree::node( T_ASSERT); t.assertion = kind; return t; }

	Tree bracket()
	{
		++m_at;
		bool negated = false;
		if (lookingAt( '^')) { negated = true; ++m_at; }
		CpRanges r;
		for (bool first=true;; first=false)
		{
			if (done()) error( "missing terminating ]");
			unsigned char c = cur();
			if (c == ']' && !first) { ++m_at; break; }
			if (c == '[' && m_at+1 < m_text.size() && m_text[ m_at+1] == ':')
			{
				size_t e = m_text.find( ":]", m_at+2);
				if (e == std::string::npos) error( "unterminated POSIX class");
				std::string nm = m_text.substr( m_at+2, e-m_at-2);
				if (nm == "alpha") { r.push_back( std::make_pair( 'A','Z')); r.push_back( std::make_pair( 'a','z')); }
				else if (nm == "digit") r.push_back( std::make_pair( '0','9'));
				else if (nm == "alnum") { r.push_back( std::make_pair( '0','9')); r.push_back( std::make_pair( 'A','Z')); r.push_back( std::make_pair( 'a','z')); }
				else if (nm == "upper") r.push_back( std::make_pair( 'A','Z'));
				else if (nm == "lower") r.push_back( std::make_pair( 'a','z'));
				else if (nm == "space") { r.push_back( std::make_pair( 9,13)); r.push_back( std::make_pair( ' ',' ')); }
				else if (nm == "punct") { r.push_back( std::make_pair( 33,47)); r.push_back( std::make_pair( 58,64)); r.push_back( std::make_pair( 91,96)); r.push_back( std::make_pair( 123,126)); }
				else if (nm == "xdigit") { r.push_back( std::make_pair( '0','9')); r.push_back( std::make_pair( 'A','F')); r.push_back( std::make_pair( 'a','f')); }
				else error( "unknown POSIX class name");
				m_at = e+2;
				continue;
			}
			uint32_t lo;
			if (c == '\\')
			{
				++m_at;
				if (done()) error( "\\ at end of pattern");
				unsigned char e = cur(); ++m_at;
				if (e=='d'||e=='w'||e=='s') { shorthand( (char)e, r); continue; }
				if (e=='D'||e=='W'||e=='S') { CpRanges t; shorthand( (char)e, t); t = complement( t); r.insert( r.end(), t.begin(), t.end()); continue; }
				if (e=='p'||e=='P') { property( e == 'P', r); continue; }
				lo = (e == 'b') ? 8 : escapedChar( e);
			}
			else lo = literalChar();
			uint32_t hi = lo;
			if (m_at+1 < m_text.size() && cur() == '-' && m_text[ m_at+1] != ']')
			{
				++m_at;
				if (cur() == '\\') { ++m_at; if (done()) error( "\\ at end of pattern"); unsigned char e = cur(); ++m_at; hi = escapedChar( e); }
				else hi = literalChar();
				if (hi < lo) error( "range out of order in character class");
			}
			r.push_back( std::make_pair( lo, hi));
		}
		fold( r);
		return fromRanges( negated ? complement( r) : r);
	}

	Tree atom()
	{
		unsigned char c = cur();
		if (c == '(')
		{
			++m_at;
			unsigned g = 0;
			if (lookingAt( '?'))
			{
				if (m_at+1 < m_text.size() && m_text[ m_at+1] == ':') m_at += 2;
				else error( "unsupported group syntax (?...)");
			}
			else g = ++m_groups;
			Tree inner = alternation();
			if (!lookingAt( ')')) error( "missing )");
			++m_at;
			Tree t = Tree::node( T_GROUP); t.group = g; t.kids.push_back( inner);
			return t;
		}
		if (c == '[') return bracket();
		if (c == '.')
		{
			++m_at;
			CpRanges r;
			if (m_dotall) r.push_back( std::make_pair( 0u, topCp()));
			else { r.push_back( std::make_pair( 0u, 9u)); r.push_back( std::make_pair( 11u, topCp())); }
			return fromRanges( r);
		}
		if (c == '^') { ++m_at; return assertNode( m_multiline ? A_BOL : A_BOD); }
		if (c == '$') { ++m_at; return assertNode( m_multiline ? A_EOL : A_EOD); }
		if (c == '*' || c == '+' || c == '?') error( "nothing to repeat");
		if (c == ')') error( "unmatched ')'");
		if (c == '\\')
		{
			++m_at;
			if (done()) error( "\\ at end of pattern");
			unsigned char e = cur(); ++m_at;
			switch (e)
			{
				case 'b': return assertNode( A_WB);
				case 'B': return assertNode( A_NWB);
				case 'A': return assertNode( A_BOD);
				case 'z': return assertNode( A_EOD);
				case 'd': case 'w': case 's': { CpRanges r; shorthand( (char)e, r); return fromRanges( r); }
				case 'D': case 'W': case 'S': { CpRanges r; shorthand( (char)e, r); return fromRanges( complement( r)); }
				case 'p': case 'P': { CpRanges r; property( e == 'P', r); return fromRanges( r); }
			}
			CpRanges r; uint32_t v = escapedChar( e); r.push_back( std::make_pair( v, v));
			fold( r);
			return fromRanges( r);
		}
		CpRanges r; uint32_t v = literalChar(); r.push_back( std::make_pair( v, v));
		fold( r);
		return fromRanges( r);
	}
};

// ---- fixed length of a subtree in bytes, or -1
int fixedLength( const Tree& t)
{
	switch (t.op)
	{
		case T_EMPTY: case T_ASSERT: return 0;
		case T_SET: return 1;
		case T_GROUP: return fixedLength( t.kids[0]);
		case T_CAT: { int s = 0; for (size_t i=0; i<t.kids.size(); ++i) { int l = fixedLength( t.kids[i]); if (l < 0) return -1; s += l; } return s; }
		case T_ALT: { int s = -2; for (size_t i=0; i<t.kids.size(); ++i) { int l = fixedLength( t.kids[i]); if (l < 0 || (s != -2 && s != l)) return -1; s = l; } return s; }
		default: return -1;
	}
}
// locate capture group g below concatenations/groups only and sum the fixed lengths around it
bool groupContext( const Tree& t, unsigned g, int& before, int& after)
{
	if (t.op == T_GROUP) return t.group == g ? true : groupContext( t.kids[0], g, before, after);
	if (t.op != T_CAT) return false;
	for (size_t i=0; i<t.kids.size(); ++i)
	{
		int b = 0, a = 0;
		if (!groupContext( t.kids[i], g, b, a)) continue;
		for (size_t k=0; k<i; ++k) { int l = fixedLength( t.kids[k]); if (l < 0) return false; b += l; }
		for (size_t k=i+1; k<t.kids.size(); ++k) { int l = fixedLength( t.kids[k]); if (l < 0) return false; a += l; }
		before += b; after += a;
		return true;
	}
	return false;
}

// ---- Glushkov construction with assertion-carrying set members
typedef std::pair<uint32_t,unsigned> Member;		// (position, assertions crossed)
struct Sets { std::vector<unsigned> nullable; std::vector<Member> first, last; };
struct Edge { uint32_t from, to; unsigned cond; };

struct Glushkov
{
	std::vector<ByteSet> positions;
	std::vector<Edge> edges;

	static void uniq( std::vector<Member>& v) { std::sort( v.begin(), v.end()); v.erase( std::unique( v.begin(), v.end()), v.end()); }
	static void uniqN( std::vector<unsigned>& v)
	{
		std::sort( v.begin(), v.end()); v.erase( std::unique( v.begin(), v.end()), v.end());
		if (!v.empty() && v[0] == 0) v.resize( 1);	// unconditional dominates
	}
	void link( const std::vector<Member>& from, const std::vector<Member>& to)
	{
		for (size_t i=0; i<from.size(); ++i) for (size_t k=0; k<to.size(); ++k)
		{
			Edge e = { from[i].first, to[k].first, from[i].second | to[k].second };
			edges.push_back( e);
		}
	}
	Sets build( const Tree& t)
	{
		Sets s;
		switch (t.op)
		{
			case T_EMPTY: s.nullable.push_back( 0); break;
			case T_ASSERT: s.nullable.push_back( t.assertion); break;
			case T_SET:
			{
				if (t.set.empty()) break;		// matches nothing
				uint32_t p = (uint32_t)positions.size(); positions.push_back( t.set);
				s.first.push_back( Member( p, 0)); s.last.push_back( Member( p, 0));
				break;
			}
			case T_GROUP: return build( t.kids[0]);
			case T_ALT:
				for (size_t i=0; i<t.kids.size(); ++i)
				{
					Sets k = build( t.kids[i]);
					s.nullable.insert( s.nullable.end(), k.nullable.begin(), k.nullable.end());
					s.first.insert( s.first.end(), k.first.begin(), k.first.end());
					s.last.insert( s.last.end(), k.last.begin(), k.last.end());
				}
				break;
			case T_CAT:
			{
				s.nullable.push_back( 0);
				for (size_t i=0; i<t.kids.size(); ++i)
				{
					Sets k = build( t.kids[i]);
					link( s.last, k.first);
					// first: extend while everything so far is nullable
					std::vector<Member> nf = s.first;
					for (size_t n=0; n<s.nullable.size(); ++n) for (size_t f=0; f<k.first.size(); ++f)
						nf.push_back( Member( k.first[f].first, k.first[f].second | s.nullable[n]));
					// last: the child's last, plus ours carried over the child's nullability
					std::vector<Member> nl = k.last;
					for (size_t n=0; n<k.nullable.size(); ++n) for (size_t l=0; l<s.last.size(); ++l)
						nl.push_back( Member( s.last[l].first, s.last[l].second | k.nullable[n]));
					std::vector<unsigned> nn;
					for (size_t a=0; a<s.nullable.size(); ++a) for (size_t b=0; b<k.nullable.size(); ++b) nn.push_back( s.nullable[a] | k.nullable[b]);
					s.first.swap( nf); s.last.swap( nl); s.nullable.swap( nn);
					uniq( s.first); uniq( s.last); uniqN( s.nullable);
				}
				break;
			}
			case T_STAR: case T_PLUS: case T_OPT:
			{
				s = build( t.kids[0]);
				if (t.op != T_OPT) link( s.last, s.first);
				if (t.op != T_PLUS) s.nullable.push_back( 0);
				break;
			}
		}
		uniq( s.first); uniq( s.last); uniqN( s.nullable);
		return s;
	}
};

// Is the expression exactly  \b <word characters> \b  or  \b ( word | word | ... ) \b  ?  Then it matches
// precisely the maximal runs of word characters that equal one of the words, and (from,to) = the run:
// no automaton bits are needed, the words go into the token hash table.
static bool literalWordOf( const Tree& t, std::string& word)
{
	word.clear();
	if (t.op == T_SET)
	{
		int only = -1, n = 0;
		for (unsigned c=0; c<256; ++c) if (t.set.has( c)) { only = (int)c; ++n; }
		if (n != 1 || !isWordChar( (unsigned)only)) return false;
		word.push_back( (char)only);
		return true;
	}
	if (t.op != T_CAT || t.kids.empty()) return false;
	for (size_t i=0; i<t.kids.size(); ++i)
	{
		std::string one;
		if (t.kids[i].op != T_SET || !literalWordOf( t.kids[i], one)) return false;
		word += one;
	}
	return !word.empty() && word.size() <= 64;
}
bool wholeWordLiterals( const Tree& t, std::vector<std::string>& words)
{
	words.clear();
	if (t.op != T_CAT || t.kids.size() < 3) return false;
	if (t.kids.front().op != T_ASSERT || t.kids.front().assertion != A_WB) return false;
	if (t.kids.back().op != T_ASSERT || t.kids.back().assertion != A_WB) return false;
	if (t.kids.size() == 3)
	{
		// \b ( w1 | w2 | .. ) \b   (the group may be capturing: the selected sub-expression is checked by the caller)
		const Tree* inner = &t.kids[1];
		while (inner->op == T_GROUP) inner = &inner->kids[0];
		if (inner->op == T_ALT)
		{
			for (size_t i=0; i<inner->kids.size(); ++i)
			{
				std::string w;
				if (!literalWordOf( inner->kids[i], w)) return false;
				if (std::find( words.begin(), words.end(), w) == words.end()) words.push_back( w);
			}
			return !words.empty();
		}
	}
	std::string word;
	for (size_t i=1; i+1<t.kids.size(); ++i)
	{
		std::string one;
		if (t.kids[i].op != T_SET || !literalWordOf( t.kids[i], one)) return false;
		word += one;
	}
	if (word.empty() || word.size() > 64) return false;
	words.push_back( word);
	return true;
}

// ---- word shapes (l1_tables.h)
struct ShapeKey { uint32_t tag, key; };
static const Tree& ungroup( const Tree& t) { const Tree* p = &t; while (p->op == T_GROUP && p->kids.size() == 1) p = &p->kids[ 0]; return *p; }
static bool singleWordByte( const Tree& t0, unsigned& b)
{
	const Tree& t = ungroup( t0);
	if (t.op != T_SET || t.set.cpRef >= 0) return false;
	int only = -1, n = 0;
	for (unsigned c=0; c<256; ++c) if (t.set.has( c)) { only = (int)c; ++n; }
	if (n != 1 || !isWordChar( (unsigned)only)) return false;
	b = (unsigned)only;
	return true;
}
static bool isSetOf( const Tree& t0, bool word)		// one byte position, every member a word character (word) / none of them one (!word)
{
	const Tree& t = ungroup( t0);
	if (t.op != T_SET || t.set.cpRef >= 0 || t.set.empty()) return false;
	for (unsigned c=0; c<256; ++c) if (t.set.has( c) && isWordChar( c) != word) return false;
	return true;
}
static bool wordAlphabetOnly( const Tree& t)		// no assertion inside, every byte position takes word characters only
{
	if (t.op == T_SET) return isSetOf( t, true);
	if (t.op == T_ASSERT) return false;
	for (size_t i=0; i<t.kids.size(); ++i) if (!wordAlphabetOnly( t.kids[ i])) return false;
	return true;
}
static unsigned minLength( const Tree& t)
{
	switch (t.op)
	{
		case T_SET: return 1;
		case T_CAT: { unsigned n = 0; for (size_t i=0; i<t.kids.size(); ++i) n += minLength( t.kids[ i]); return n; }
		case T_ALT: { unsigned n = ~0u; for (size_t i=0; i<t.kids.size(); ++i) { unsigned k = minLength( t.kids[ i]); if (k < n) n = k; } return t.kids.empty() ? 0 : n; }
		case T_PLUS: case T_GROUP: return t.kids.empty() ? 0 : minLength( t.kids[ 0]);
		default: return 0;
	}
}
static bool isWordBoundary( const Tree& t) { return t.op == T_ASSERT && t.assertion == A_WB; }
static uint32_t polyHashOf( const std::string& w)
{
	uint32_t h = 0;
	for (size_t i=0; i<w.size(); ++i) h = h * (uint32_t)L1_LITHASH_MUL + (uint32_t)(unsigned char)w[ i] + 1u;
	return literalHashFinish( h);
}
// Is the expression a word shape?  Completeness of the candidates the lexer derives from the key (every match is among them):
//  SUFFIX   X b1..bk \b, the b word characters: the match's last byte is a word character and \b holds behind it, so the match
//           ends where a run of word characters ends, and the run's last k bytes are b1..bk (they are adjacent word characters).
//  PREFIX   \b S1..So b1..bk E \b with every byte position of S, b, E taking word characters only and no assertion inside: the
//           match consists of word characters, starts where a run starts (\b before a word character) and ends where it ends:
//           it IS the run, whose bytes [o, o+k) are b1..bk.
//  PREVWORD \b b1..bn SEP E \b, SEP one position without word characters, E as above and not empty: b1..bn is a whole run
//           (\b before it, SEP behind it), E the whole next run, one byte apart; the match ends where that run ends.
static bool wordShapeOf( const Tree& t, ShapeKey& out)
{
	if (t.op != T_CAT || t.kids.size() < 3) return false;
	const size_t n = t.kids.size();
	if (!isWordBoundary( t.kids[ n-1])) return false;
	{
		size_t k = 0; unsigned b;
		while (k < n-1 && singleWordByte( t.kids[ n-2-k], b)) ++k;
		if (k >= 2 && k < n-1)
		{
			const size_t kk = k > 4 ? 4 : k;
			uint32_t key = 0;
			for (size_t i=0; i<kk; ++i) { singleWordByte( t.kids[ n-1-kk+i], b); key |= (uint32_t)b << (8*i); }
			out.tag = (uint32_t)SHAPE_SUFFIX | ((uint32_t)kk << 4); out.key = key;
			return true;
		}
	}
	if (!isWordBoundary( t.kids[ 0])) return false;
	bool allWord = true;
	for (size_t i=1; i+1<n && allWord; ++i) allWord = wordAlphabetOnly( t.kids[ i]);
	if (allWord)
	{
		for (size_t o=0; o<=3 && 1+o+2 <= n-1; ++o)
		{
			if (o && !isSetOf( t.kids[ o], true)) break;		// S1..So: single positions
			size_t k = 0; unsigned b;
			while (1+o+k < n-1 && singleWordByte( t.kids[ 1+o+k], b)) ++k;
			if (k >= 2)
			{
				const size_t kk = k > 4 ? 4 : k;
				uint32_t key = 0;
				for (size_t i=0; i<kk; ++i) { singleWordByte( t.kids[ 1+o+i], b); key |= (uint32_t)b << (8*i); }
				out.tag = (uint32_t)SHAPE_PREFIX | ((uint32_t)o << 2) | ((uint32_t)kk << 4); out.key = key;
				return true;
			}
		}
		return false;
	}
	{
		std::string word; unsigned b;
		size_t i = 1;
		while (i+1 < n && singleWordByte( t.kids[ i], b)) { word.push_back( (char)b); ++i; }
		if (word.empty() || word.size() > 64 || i+1 >= n || !isSetOf( t.kids[ i], false)) return false;
		++i;
		if (i+1 > n-1) return false;
		unsigned restMin = 0;
		for (size_t r=i; r+1<n; ++r) { if (!wordAlphabetOnly( t.kids[ r])) return false; restMin += minLength( t.kids[ r]); }
		if (restMin < 1) return false;
		out.tag = (uint32_t)SHAPE_PREVWORD | ((uint32_t)word.size() << 8); out.key = polyHashOf( word);
		return true;
	}
}

int ctxOfByte( unsigned c) { return c == '\n' ? CTX_NEWLINE : isWordChar( c) ? CTX_WORD : CTX_OTHER; }
bool ctxIsWord( int ctx) { return ctx == CTX_WORD; }
// does the conjunction `cond` hold between a byte of context `prev` and one of context `next`?
bool condHolds( unsigned cond, int prev, int next)
{
	if ((cond & A_WB) && ctxIsWord( prev) == ctxIsWord( next)) return false;
	if ((cond & A_NWB) && ctxIsWord( prev) != ctxIsWord( next)) return false;
	if ((cond & A_SAMEWORD) && ctxIsWord( prev) != ctxIsWord( next)) return false;
	if ((cond & A_BOL) && !(prev == CTX_EDGE || prev == CTX_NEWLINE)) return false;
	if ((cond & A_EOL) && !(next == CTX_EDGE || next == CTX_NEWLINE)) return false;
	if ((cond & A_BOD) && prev != CTX_EDGE) return false;
	if ((cond & A_EOD) && next != CTX_EDGE) return false;
	return true;
}

// one compiled pattern in its own bit space (bit i = position i)
struct Automaton
{
	std::vector<ByteSet> pos;		// byte set per position
	std::vector<int> ctxDef;		// definite context of the position (it was split by context because it touches an assertion), or -1
	std::vector<uint64_t> follow;		// follow[i] = successors of position i
	uint64_t start[ CTX_COUNT];		// positions a match may begin with, by context of the byte before
	uint64_t accept[ CTX_COUNT];		// positions a match may end with, by context of the byte after
	uint32_t emptyOk;			// bit (prev*CTX_COUNT + next): the expression matches the empty string between a byte of context prev and one of context next
};

struct TooWide {};		// more than 64 byte positions: the caller may cut the expression at an alternation

Automaton makeAutomaton( const Tree& tree, const std::string&, bool ucp)
{
	Glushkov g;
	Sets root = g.build( tree);
	const size_t n0 = g.positions.size();
	// which positions touch an assertion
	std::vector<char> touched( n0, 0);
	for (size_t i=0; i<g.edges.size(); ++i) if (g.edges[i].cond) { touched[ g.edges[i].from] = 1; touched[ g.edges[i].to] = 1; }
	for (size_t i=0; i<root.first.size(); ++i) if (root.first[i].second) touched[ root.first[i].first] = 1;
	for (size_t i=0; i<root.last.size(); ++i) if (root.last[i].second) touched[ root.last[i].first] = 1;
	// split touched positions by context class so that their context is definite
	Automaton a;
	std::vector<std::vector<uint32_t> > parts( n0);	// old position -> new positions
	std::vector<int> ctxOf;				// new position -> definite context or -1
	for (size_t p=0; p<n0; ++p)
	{
		if (!touched[ p])
		{
			parts[ p].push_back( (uint32_t)a.pos.size()); a.pos.push_back( g.positions[ p]); ctxOf.push_back( -1);
			continue;
		}
		ByteSet byCtx[3];
		for (unsigned c=0; c<256; ++c) if (g.positions[ p].has( c))
		{
			byCtx[ ctxOfByte( c)].add( c);
			// UCP: a byte beyond ASCII belongs to a word character or to another one -- which, the position's class says
			// at run time (classes by code point for lead bytes, twin classes for continuation bytes)
			if (ucp && c >= 0x80) byCtx[ CTX_WORD].add( c);
		}
		for (int k=0; k<3; ++k) if (!byCtx[k].empty())
		{
			byCtx[k].cpRef = g.positions[ p].cpRef;
			parts[ p].push_back( (uint32_t)a.pos.size()); a.pos.push_back( byCtx[k]); ctxOf.push_back( k);
		}
	}
	if (a.pos.size() > 64) throw TooWide();
	a.ctxDef = ctxOf;
	a.emptyOk = 0;
	for (size_t i=0; i<root.nullable.size(); ++i) for (int pv=0; pv<CTX_COUNT; ++pv) for (int nx=0; nx<CTX_COUNT; ++nx)
	{
		if (condHolds( root.nullable[ i], pv, nx)) a.emptyOk |= 1u << (pv*CTX_COUNT + nx);
	}
	a.follow.assign( a.pos.size(), 0);
	for (int c=0; c<CTX_COUNT; ++c) { a.start[c] = 0; a.accept[c] = 0; }
	for (size_t i=0; i<g.edges.size(); ++i)
	{
		const Edge& e = g.edges[i];
		for (size_t x=0; x<parts[ e.from].size(); ++x) for (size_t y=0; y<parts[ e.to].size(); ++y)
		{
			uint32_t pf = parts[ e.from][x], pt = parts[ e.to][y];
			if (e.cond == 0 || condHolds( e.cond, ctxOf[ pf], ctxOf[ pt])) a.follow[ pf] |= (1ull << pt);
		}
	}
	for (size_t i=0; i<root.first.size(); ++i)
	{
		const std::vector<uint32_t>& ps = parts[ root.first[i].first];
		for (size_t x=0; x<ps.size(); ++x) for (int prev=0; prev<CTX_COUNT; ++prev)
		{
			if (root.first[i].second == 0 || condHolds( root.first[i].second, prev, ctxOf[ ps[x]])) a.start[ prev] |= (1ull << ps[x]);
		}
	}
	for (size_t i=0; i<root.last.size(); ++i)
	{
		const std::vector<uint32_t>& ps = parts[ root.last[i].first];
		for (size_t x=0; x<ps.size(); ++x) for (int next=0; next<CTX_COUNT; ++next)
		{
			if (root.last[i].second == 0 || condHolds( root.last[i].second, ctxOf[ ps[x]], next)) a.accept[ next] |= (1ull << ps[x]);
		}
	}
	return a;
}

size_t leafCount( const Tree& t)
{
	if (t.op == T_SET) return 1;
	size_t n = 0;
	for (size_t i=0; i<t.kids.size(); ++i) n += leafCount( t.kids[ i]);
	return n;
}
// the widest alternation that is reached through concatenations and groups only: prefix (a|b|..) suffix is the
// union of prefix (a|..) suffix and prefix (..|z) suffix
Tree* widestAlternation( Tree& t)
{
	if (t.op == T_ALT) return &t;
	if (t.op == T_GROUP) return widestAlternation( t.kids[ 0]);
	if (t.op != T_CAT) return 0;
	Tree* best = 0; size_t bestN = 0;
	for (size_t i=0; i<t.kids.size(); ++i)
	{
		Tree* c = widestAlternation( t.kids[ i]);
		if (c && leafCount( *c) > bestN) { best = c; bestN = leafCount( *c); }
	}
	return best;
}
// automata of an expression: one, or several that accept its language between them
void makeAutomata( const Tree& tree, const std::string& expr, bool ucp, std::vector<Automaton>& out, unsigned depth=0)
{
	try { out.push_back( makeAutomaton( tree, expr, ucp)); return; }
	catch (const TooWide&) {}
	Tree lo = tree, hi = tree;
	Tree* a = widestAlternation( lo);
	Tree* b = widestAlternation( hi);
	if (!a || a->kids.size() < 2 || depth > 8)
	{
		throw std::runtime_error( "failed to compile pattern \"" + expr + "\": expression too complex (more than 64 byte positions outside of an alternation)");
	}
	// halves of about equal width
	const size_t total = leafCount( *a);
	size_t cut = 0, acc = 0;
	while (cut + 1 < a->kids.size() && acc + leafCount( a->kids[ cut]) <= total/2) { acc += leafCount( a->kids[ cut]); ++cut; }
	if (cut == 0) cut = 1;
	std::vector<Tree> first( a->kids.begin(), a->kids.begin() + cut), second( a->kids.begin() + cut, a->kids.end());
	*a = Tree::alt( first); *b = Tree::alt( second);
	makeAutomata( lo, expr, ucp, out, depth+1);
	makeAutomata( hi, expr, ucp, out, depth+1);
	if (out.size() > 32) throw std::runtime_error( "failed to compile pattern \"" + expr + "\": expression too complex (more than 32 words of 64 byte positions)");
}

bool hasWordAssertion( const Tree& t)
{
	if (t.op == T_ASSERT && (t.assertion & (A_WB|A_NWB))) return true;
	for (size_t i=0; i<t.kids.size(); ++i) if (hasWordAssertion( t.kids[ i])) return true;
	return false;
}
// UCP: is the code point a word character (\\w = L | N | _)?
bool ucpWordCp( uint32_t cp)
{
	if (cp < 0x80) return isWordChar( cp);
	static const char* cats[2] = {"L", "N"};
	for (int ci=0; ci<2; ++ci) for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
	{
		if (std::strcmp( c->name, cats[ ci])) continue;
		size_t lo = 0, hi = c->count;
		while (lo < hi) { size_t mid = (lo+hi)/2; if (c->ranges[ mid].hi < cp) lo = mid+1; else hi = mid; }
		if (lo < c->count && c->ranges[ lo].lo <= cp) return true;
	}
	return false;
}

bool sameNoCase( const std::string& a, const char* b)
{
	size_t n = std::strlen( b);
	if (a.size() != n) return false;
	for (size_t i=0; i<n; ++i) { char x = a[i], y = b[i]; if (x>='a'&&x<='z') x -= 32; if (y>='a'&&y<='z') y -= 32; if (x != y) return false; }
	return true;
}

// src/patternLexer.cpp:605-626: a trailing "~<digits>" is the edit distance
unsigned cutEditDistance( std::string& expr)
{
	size_t e = expr.size();
	while (e > 0 && (unsigned char)expr[e-1] <= 32) --e;
	size_t d = e;
	while (d > 0 && expr[d-1] >= '0' && expr[d-1] <= '9') --d;
	if (d == e) return 0;
	size_t t = d;
	while (t > 0 && (unsigned char)expr[t-1] <= 32) --t;
	if (t == 0 || expr[t-1] != '~') return 0;
	unsigned dist = (unsigned)atoi( expr.c_str() + d);
	--t;
	while (t > 0 && (unsigned char)expr[t-1] <= 32) --t;
	expr.resize( t);
	return dist;
}

} // namespace

// src/patternLexer.cpp:971-988
void LexCompiler::defineLexemName( uint32_t id, const std::string& name)
{
	if (m_names.count( id)) throw std::runtime_error( "duplicate definition");
	m_names[ id] = name;
}
const char* LexCompiler::getLexemName( uint32_t id) const
{
	std::map<uint32_t,std::string>::const_iterator it = m_names.find( id);
	return it == m_names.end() ? 0 : it->second.c_str();
}

// src/patternLexer.cpp:990-1006, :245-262, limits :76-99
void LexCompiler::defineLexem( uint32_t id, const std::string& expression, uint32_t resultIndex, uint32_t level, int posbind)
{
	if (m_compiled) throw std::runtime_error( "called define pattern after calling 'compile'");
	if (id > (1u<<30)-1) throw std::runtime_error( "pattern id out of range, It must be a positive integer in the range 1..1073741823");
	if (level > 255) throw std::runtime_error( "level out of range, It must be a positive integer in the range 1..255");
	if (resultIndex > 255) throw std::runtime_error( "result index out of range, It must be a positive integer in the range 1..255");
	if (posbind < 0 || posbind > 3) throw std::runtime_error( "unknown position bind value");
	Def d; d.expression = expression; d.editdist = cutEditDistance( d.expression);
	if (d.editdist > 255) throw std::runtime_error( "edit distance out of range, It must be a positive integer in the range 1..255");
	d.id = id; d.resultIndex = resultIndex; d.level = level; d.posbind = posbind;
	m_defs.push_back( d);
}

// src/patternLexer.cpp:1008-1019, :264-293
void LexCompiler::defineSymbol( uint32_t symbolid, uint32_t patternid, const std::string& name)
{
	if (m_compiled) throw std::runtime_error( "called define pattern after calling 'compile'");
	if (patternid > (1u<<30)-1 || symbolid > (1u<<30)-1) throw std::runtime_error( "symbol or pattern id out of range");
	std::map<std::string,uint32_t>& tab = m_symbols[ patternid];
	if (m_symbols.size() > 255) throw std::runtime_error( "too many pattern symbol tables defined");	// :440-445
	if (tab.count( name)) throw std::runtime_error( "symbol defined twice: '" + name + "'");
	tab[ name] = symbolid;
}

// src/patternLexer.cpp:1020-1029, :295-310
uint32_t LexCompiler::getSymbol( uint32_t patternid, const std::string& name) const
{
	std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator t = m_symbols.find( patternid);
	if (t == m_symbols.end()) return 0;
	std::map<std::string,uint32_t>::const_iterator s = t->second.find( name);
	return s == t->second.end() ? 0 : s->second;
}

// src/patternLexer.cpp:1031-1066
void LexCompiler::defineOption( const std::string& name, double)
{
	if (sameNoCase( name, "CASELESS")) m_options |= LEX_CASELESS;
	else if (sameNoCase( name, "DOTALL")) m_options |= LEX_DOTALL;
	else if (sameNoCase( name, "MULTILINE")) m_options |= LEX_MULTILINE;
	else if (sameNoCase( name, "ALLOWEMPTY")) m_options |= LEX_ALLOWEMPTY;
	else if (sameNoCase( name, "UCP")) m_options |= LEX_UCP;
	else if (sameNoCase( name, "BYTECHAR")) m_options |= LEX_BYTECHAR;
	else throw std::runtime_error( "unknown option '" + name + "'");
}

// src/patternLexer.cpp:1068-1118 (+ PatternTable::complete :333-412)
void LexCompiler::compile()
{
	LexTables& T = m_tables;
	T = LexTables();

	// 0. a table with an edit distance expression: every expression takes the approximate route
	//    (src/patternLexer.cpp:333-352); supported for tables of plain literals
	bool approxTable = false;
	for (size_t di=0; di<m_defs.size(); ++di) if (m_defs[ di].editdist) approxTable = true;
	if (m_options & LEX_BYTECHAR) approxTable = true;		// forceOneByteCharMap (:1055-1058): the same route without an edit distance
	if (approxTable)
	{
		if (m_options & LEX_CASELESS) throw std::runtime_error( "approximate matching (~N) together with CASELESS is not supported by this lexer");
		if (!m_symbols.empty()) throw std::runtime_error( "approximate matching (~N) together with symbols is not supported by this lexer");
		if (m_defs.size() > L1_APPROX_MAXPATTERNS) throw std::runtime_error( "too many expressions in a table with approximate matching (~N)");
	}

	// 1. per pattern automata
	const bool ucp = (m_options & LEX_UCP) != 0;
	if (ucp && approxTable) throw std::runtime_error( "approximate matching (~N) together with UCP is not supported by this lexer");
	std::vector<CpRanges> cpSets;			// code point sets of the leaves with ByteSet::cpRef
	std::vector<Automaton> autos;
	std::map<std::string,std::vector<uint32_t> > literalWords;
	std::vector<std::pair<uint32_t,ShapeKey> > shapeOf;		// (patterns[] entry, key) of the expressions that are word shapes
	T.patterns.clear();
	for (size_t di=0; di<m_defs.size(); ++di)
	{
		const Def& d = m_defs[ di];
		if (approxTable)
		{
			if (d.expression.empty() || d.expression.find_first_of( "\\.[](){}|*+?^$") != std::string::npos || d.resultIndex)
			{
				throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": a table with approximate matching (~N) or option BYTECHAR holds plain literal expressions only in this lexer");
			}
			DevApproxPattern ap; std::memset( &ap, 0, sizeof(ap));
			ap.id = d.id; ap.levelBind = (d.level & 0xFF) | ((uint32_t)d.posbind << 8); ap.editdist = d.editdist; ap.byteLen = (uint32_t)d.expression.size();
			const unsigned char* s = (const unsigned char*)d.expression.data();
			for (size_t at=0; at<d.expression.size();)
			{
				// lenient UTF-8 (the kernel's rule): a lead byte with all its continuation bytes, else the byte itself
				unsigned char c = s[ at];
				unsigned want = (c >= 0xC2 && c <= 0xDF) ? 2 : (c >= 0xE0 && c <= 0xEF) ? 3 : (c >= 0xF0 && c <= 0xF4) ? 4 : 1;
				uint32_t cp = c; unsigned n = 1;
				if (want > 1 && at + want <= d.expression.size())
				{
					uint32_t v = c & (0xFFu >> (want+1)); bool ok = true;
					for (unsigned i=1; i<want; ++i) { if ((s[ at+i] & 0xC0) != 0x80) { ok = false; break; } v = (v << 6) | (s[ at+i] & 0x3F); }
					if (ok) { cp = v; n = want; }
				}
				if (ap.len >= L1_APPROX_MAXCHARS) throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": a literal with approximate matching has at most 24 characters");
				ap.cp[ ap.len++] = cp; at += n;
			}
			if (d.editdist > L1_APPROX_MAXDIST || d.editdist >= ap.len) throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": the edit distance is at most 3 and below the number of characters");
			T.approx.push_back( ap);
			DevLexPattern dp; std::memset( &dp, 0, sizeof(dp));
			dp.id = d.id; dp.levelBind = ap.levelBind; dp.word = L1_WORD_LITERAL; dp.defIndex = (uint32_t)di;
			T.patterns.push_back( dp);
			autos.push_back( Automaton());
			for (int c=0; c<CTX_COUNT; ++c) { autos.back().start[c] = 0; autos.back().accept[c] = 0; }
			continue;
		}
		const size_t cpSetsBefore = cpSets.size();
		Syntax syn( d.expression, m_options, &cpSets, false);
		Tree tree = syn.run();
		if (ucp && hasWordAssertion( tree))
		{
			// \\b / \\B by Unicode word characters: the bytes of one character have to agree on it
			cpSets.resize( cpSetsBefore);
			Syntax again( d.expression, m_options, &cpSets, true);
			tree = again.run();
		}
		DevLexPattern dp; std::memset( &dp, 0, sizeof(dp));
		dp.id = d.id; dp.defIndex = (uint32_t)di;
		dp.levelBind = (d.level & 0xFF) | ((uint32_t)d.posbind << 8);
		if (d.resultIndex)
		{
			int before = 0, after = 0;
			if (d.resultIndex > syn.groups() || !groupContext( tree, d.resultIndex, before, after))
			{
				throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": selecting a sub expression needs fixed-length context around the group in this lexer");
			}
			dp.prefixLen = (uint32_t)before; dp.suffixLen = (uint32_t)after;
			dp.levelBind |= (1u << 17);
		}
		if (m_symbols.count( d.id)) dp.levelBind |= (1u << 16);
		std::vector<std::string> words;
		if (!d.resultIndex && wholeWordLiterals( tree, words))
		{
			dp.word = L1_WORD_LITERAL;
			for (size_t wi=0; wi<words.size(); ++wi) literalWords[ words[ wi]].push_back( (uint32_t)T.patterns.size());
			T.patterns.push_back( dp);
			autos.push_back( Automaton());
			for (int c=0; c<CTX_COUNT; ++c) { autos.back().start[c] = 0; autos.back().accept[c] = 0; }
			continue;
		}
		std::vector<Automaton> parts;
		makeAutomata( tree, d.expression, ucp, parts);
		if (!(m_options & LEX_ALLOWEMPTY) && (parts[ 0].emptyOk & (1u << (CTX_EDGE*CTX_COUNT + CTX_EDGE))))
		{
			// what Hyperscan answers hs_compile_ext_multi with (the reference hands the message on, patternLexer.cpp:1094-1104)
			throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": Pattern matches empty buffer; use option ALLOWEMPTY to enable support");
		}
		if ((m_options & LEX_ALLOWEMPTY) && parts[ 0].emptyOk)
		{
			// HS_FLAG_ALLOWEMPTY: the expression also reports its empty matches (one report per offset where nothing longer ends)
			if (parts.size() > 1) throw std::runtime_error( "failed to compile pattern \"" + d.expression + "\": an expression that matches the empty string must fit one automaton word with ALLOWEMPTY");
			DevNullable nl; nl.pattern = (uint32_t)T.patterns.size(); nl.emptyOk = parts[ 0].emptyOk; nl._pad[0] = 0; nl._pad[1] = 0;
			T.nullable.push_back( nl);
		}
		{
			ShapeKey sk;
			if (parts.size() == 1 && !parts[ 0].emptyOk && wordShapeOf( tree, sk)) shapeOf.push_back( std::make_pair( (uint32_t)T.patterns.size(), sk));
		}
		for (size_t k=0; k<parts.size(); ++k) { T.patterns.push_back( dp); autos.push_back( parts[ k]); }
	}
	// word shapes are taken when the table is a plain one: words by ASCII word characters (no UCP), no classes by code point, the
	// default layout (SPA_L1_SHAPES=0 keeps every expression in the scanned passes: tests)
	std::vector<char> isShape( autos.size(), 0);
	{
		const char* sw = getenv( "SPA_L1_SHAPES");
		const bool off = (sw && sw[ 0] == '0') || getenv( "SPA_L1_SHARE") || ucp || !cpSets.empty() || approxTable || (m_options & LEX_ALLOWEMPTY);
		if (off) shapeOf.clear();
		// the lexer probes one key per variant (kind, place, length) at every end of a word: the most populated ones are kept
		std::map<uint32_t,size_t> population;
		auto variantOf = []( const ShapeKey& k) -> uint32_t { return (k.tag & 3u) == (uint32_t)SHAPE_PREVWORD ? (uint32_t)SHAPE_PREVWORD : k.tag; };
		for (size_t i=0; i<shapeOf.size(); ++i) population[ variantOf( shapeOf[ i].second)] += 1;
		if (population.size() > SHAPE_MAXVARIANTS)
		{
			std::vector<std::pair<size_t,uint32_t> > byPop;
			for (std::map<uint32_t,size_t>::const_iterator pi=population.begin(); pi!=population.end(); ++pi) byPop.push_back( std::make_pair( pi->second, pi->first));
			std::sort( byPop.begin(), byPop.end());
			std::set<uint32_t> dropped;
			for (size_t i=0; i+SHAPE_MAXVARIANTS<byPop.size(); ++i) dropped.insert( byPop[ i].second);
			std::vector<std::pair<uint32_t,ShapeKey> > kept;
			for (size_t i=0; i<shapeOf.size(); ++i) if (!dropped.count( variantOf( shapeOf[ i].second))) kept.push_back( shapeOf[ i]);
			shapeOf.swap( kept);
		}
		for (size_t i=0; i<shapeOf.size(); ++i) isShape[ shapeOf[ i].first] = 1;
	}
	if (T.patterns.size() >= (1u << 24)) throw std::runtime_error( "too many patterns");

	// 2. layout: a pattern never straddles a 64-bit word.  Patterns in definition order make the report
	//    order for equal end offsets (ascending pattern index) fall out of the (pass, lane, bit) order;
	//    but in-order packing leaves ~8% of the bits unused, and when that costs a whole pass (every pass
	//    is a full unit of work per input byte) the patterns are packed first-fit by decreasing size
	//    instead and the kernel sorts the (few) reports that share an end offset.
	std::vector<uint32_t> bitBase( autos.size(), 0);
	std::vector<uint32_t> wordOf( autos.size(), 0);
	T.nofPositions = 0;
	uint32_t word = 0;
	{
		uint32_t used = 0; bool any = false;
		for (size_t pi=0; pi<autos.size(); ++pi)
		{
			uint32_t n = (uint32_t)autos[ pi].pos.size();
			if (T.patterns[ pi].word == L1_WORD_LITERAL || isShape[ pi]) continue;
			T.nofPositions += n;
			if (used + n > 64) { ++word; used = 0; }
			bitBase[ pi] = used; wordOf[ pi] = word; used += n; any = true;
		}
		if (!any) word = 0; else ++word;		// word = number of words used
	}
	T.reportsOrdered = true;
	{
		const uint32_t perPass = L1_WORDS_PER_PASS;
		const uint32_t passesInOrder = (word + perPass-1) / perPass;
		const uint32_t passesMin = (T.nofPositions + 64*perPass-1) / (64*perPass);
		if (passesInOrder > passesMin && passesInOrder > 1)
		{
			std::vector<size_t> bySize;
			for (size_t pi=0; pi<autos.size(); ++pi) if (T.patterns[ pi].word != L1_WORD_LITERAL && !isShape[ pi]) bySize.push_back( pi);
			std::stable_sort( bySize.begin(), bySize.end(), [&]( size_t a, size_t b) { return autos[ a].pos.size() > autos[ b].pos.size(); });
			std::vector<uint32_t> fill;
			std::vector<uint32_t> base2( autos.size(), 0), word2( autos.size(), 0);
			size_t firstOpen = 0;
			for (size_t k=0; k<bySize.size(); ++k)
			{
				const size_t pi = bySize[ k];
				const uint32_t n = (uint32_t)autos[ pi].pos.size();
				size_t wi = firstOpen;
				while (wi < fill.size() && fill[ wi] + n > 64) ++wi;
				if (wi == fill.size()) fill.push_back( 0);
				base2[ pi] = fill[ wi]; word2[ pi] = (uint32_t)wi; fill[ wi] += n;
				while (firstOpen < fill.size() && fill[ firstOpen] == 64) ++firstOpen;
			}
			const uint32_t passesPacked = ((uint32_t)fill.size() + perPass-1) / perPass;
			if (passesPacked < passesInOrder)
			{
				bitBase.swap( base2); wordOf.swap( word2); word = (uint32_t)fill.size();
				T.reportsOrdered = false;
			}
		}
	}
	// the passes the scan kernel runs end here; the word shapes follow in passes of their own (in definition order)
	T.scanPasses = (word + L1_WORDS_PER_PASS-1) / L1_WORDS_PER_PASS;
	T.scanWords = word;
	T.lanesOk = true;
	for (size_t pi=0; pi<autos.size() && T.lanesOk; ++pi)
	{
		if (T.patterns[ pi].word == L1_WORD_LITERAL || isShape[ pi]) continue;
		const Automaton& a = autos[ pi];
		const size_t n = a.pos.size();
		for (size_t k=0; k<n && T.lanesOk; ++k)
		{
			if (!a.pos[ k].has( ' ')) continue;
			// is position k on a cycle?  (reachability over the follow edges, at most 64 positions)
			uint64_t seen = 0, front = a.follow[ k];
			while (front & ~seen)
			{
				const uint64_t fresh = front & ~seen;
				seen |= fresh; front = 0;
				for (size_t q=0; q<n; ++q) if (fresh & (1ull << q)) front |= a.follow[ q];
			}
			if (seen & (1ull << k)) T.lanesOk = false;
		}
	}
	T.nofShapes = (uint32_t)shapeOf.size();
	if (!shapeOf.empty())
	{
		uint32_t w2 = T.scanPasses * L1_WORDS_PER_PASS, used = 0;
		for (size_t pi=0; pi<autos.size(); ++pi)
		{
			if (!isShape[ pi]) continue;
			const uint32_t n = (uint32_t)autos[ pi].pos.size();
			T.nofPositions += n;
			if (used + n > 64) { ++w2; used = 0; }
			bitBase[ pi] = used; wordOf[ pi] = w2; used += n;
		}
		word = w2 + 1;
	}
	// bit of every position of every pattern inside its word (contiguous in the two layouts above)
	std::vector<std::vector<uint8_t> > bitOf( autos.size());
	for (size_t pi=0; pi<autos.size(); ++pi)
	{
		if (T.patterns[ pi].word == L1_WORD_LITERAL) continue;
		for (size_t k=0; k<autos[ pi].pos.size(); ++k) bitOf[ pi].push_back( (uint8_t)(bitBase[ pi] + k));
	}
	// 2b. shared first position.  Many sets hold families of patterns that begin alike ([a-z]+ing\b,
	//     [a-z]+ed\b, ..; \bun\w+, \bup\w+, ..).  When the first position of a pattern is its only start
	//     position, is not accepting and has no edge coming back into it, "position 0 is live" means the
	//     same for every pattern with an identical position 0 (same bytes, same self loop, same start
	//     contexts): such patterns can share that one bit when they sit in the same word -- the edges
	//     from the shared bit to each pattern's own positions become one exception row of that word, and
	//     the pattern's mask (accept attribution, start-of-match run) is its own bits plus the shared one.
	{
		// Opt-in (SPA_L1_SHARE=on: when it saves a pass, =force: always; tests): on the 10k-pattern benchmark set
		// it takes the tables from 3 passes to 2 but the lexer kernel only from 226 to 219 ms -- the per-byte
		// scalar work and the report handling dominate, not the pass count -- so the default stays the plain layout.
		const char* shareEnv = getenv( "SPA_L1_SHARE");
		const bool shareForce = shareEnv && !std::strcmp( shareEnv, "force");
		const bool shareOff = !(shareForce || (shareEnv && !std::strcmp( shareEnv, "on")));
		std::map<std::string,std::vector<size_t> > groups;
		std::vector<size_t> singles;
		for (size_t pi=0; pi<autos.size() && !shareOff; ++pi)
		{
			if (T.patterns[ pi].word == L1_WORD_LITERAL) continue;
			const Automaton& a = autos[ pi];
			bool ok = a.pos.size() >= 2;
			uint64_t anyStart = 0;
			for (int c=0; c<CTX_COUNT && ok; ++c) { ok = (a.start[ c] & ~1ull) == 0 && (a.accept[ c] & 1ull) == 0; anyStart |= a.start[ c]; }
			for (size_t k=1; k<a.pos.size() && ok; ++k) ok = (a.follow[ k] & 1ull) == 0;
			if (!ok || !anyStart) { singles.push_back( pi); continue; }
			std::string key;
			for (unsigned c=0; c<256; c+=8) { unsigned char b = 0; for (unsigned x=0; x<8; ++x) if (a.pos[ 0].has( c+x)) b |= (unsigned char)(1u << x); key.push_back( (char)b); }
			key.push_back( (char)(a.follow[ 0] & 1ull)); key.push_back( (char)(a.pos[ 0].cpRef & 0xFF)); key.push_back( (char)((a.pos[ 0].cpRef >> 8) & 0xFF));
			for (int c=0; c<CTX_COUNT; ++c) key.push_back( (char)(a.start[ c] & 1ull));
			groups[ key].push_back( pi);
		}
		if (!shareOff)
		{
			struct Bin { uint32_t used; std::vector<size_t> members; bool shared; };
			std::vector<Bin> bins;
			for (std::map<std::string,std::vector<size_t> >::iterator gi=groups.begin(); gi!=groups.end(); ++gi)
			{
				std::vector<size_t>& mem = gi->second;
				if (mem.size() < 2) { singles.push_back( mem[ 0]); continue; }
				std::stable_sort( mem.begin(), mem.end(), [&]( size_t a, size_t b) { return autos[ a].pos.size() > autos[ b].pos.size(); });
				const size_t firstBin = bins.size();
				for (size_t k=0; k<mem.size(); ++k)
				{
					const uint32_t chain = (uint32_t)autos[ mem[ k]].pos.size() - 1;
					size_t bi = firstBin;
					while (bi < bins.size() && bins[ bi].used + chain > 64) ++bi;
					if (bi == bins.size()) { Bin b; b.used = 1; b.shared = true; bins.push_back( b); }
					bins[ bi].members.push_back( mem[ k]); bins[ bi].used += chain;
				}
			}
			// everything else first-fit by decreasing size into what is left (a bin opened here has no shared bit)
			std::sort( singles.begin(), singles.end());
			std::stable_sort( singles.begin(), singles.end(), [&]( size_t a, size_t b) { return autos[ a].pos.size() > autos[ b].pos.size(); });
			for (size_t k=0; k<singles.size(); ++k)
			{
				const Automaton& a = autos[ singles[ k]];
				const uint32_t n = (uint32_t)a.pos.size();
				// a pattern with exception edges of its own stays out of the words that spend their exception
				// row on a shared first position (a second row is paid by every word of the pass)
				bool ownEx = false;
				for (uint32_t q=0; q<n && !ownEx; ++q) ownEx = (a.follow[ q] & ~(1ull << q) & ~(q+1 < n ? (1ull << (q+1)) : 0ull)) != 0;
				size_t bi = 0;
				while (bi < bins.size() && (bins[ bi].used + n > 64 || (ownEx && bins[ bi].shared))) ++bi;
				if (bi == bins.size()) { Bin b; b.used = 0; b.shared = false; bins.push_back( b); }
				bins[ bi].members.push_back( singles[ k] | ((size_t)1 << 62)); bins[ bi].used += n;
			}
			const uint32_t perPass = L1_WORDS_PER_PASS;
			const uint32_t passesNow = word ? (word + perPass-1) / perPass : 1;
			const uint32_t passesShared = bins.empty() ? 1 : ((uint32_t)bins.size() + perPass-1) / perPass;
			if (!bins.empty() && (passesShared < passesNow || shareForce))
			{
				T.nofPositions = 0;
				for (size_t bi=0; bi<bins.size(); ++bi)
				{
					uint32_t next = bins[ bi].shared ? 1 : 0;
					T.nofPositions += next;
					for (size_t k=0; k<bins[ bi].members.size(); ++k)
					{
						const bool single = (bins[ bi].members[ k] >> 62) != 0;
						const size_t pi = bins[ bi].members[ k] & (((size_t)1 << 62) - 1);
						const size_t n = autos[ pi].pos.size();
						bitOf[ pi].clear();
						wordOf[ pi] = (uint32_t)bi;
						if (!single) bitOf[ pi].push_back( 0);			// the shared first position
						for (size_t q=single?0:1; q<n; ++q) { bitOf[ pi].push_back( (uint8_t)next++); ++T.nofPositions; }
					}
				}
				word = (uint32_t)bins.size();
				T.reportsOrdered = false;
			}
		}
	}
	T.wordPatBegin.clear(); T.wordPats.clear();
	{
		// patterns of every word (ascending pattern index inside a word)
		std::vector<std::vector<uint32_t> > perWord( word);
		for (size_t pi=0; pi<autos.size(); ++pi)
		{
			if (T.patterns[ pi].word == L1_WORD_LITERAL) continue;
			T.patterns[ pi].word = wordOf[ pi];
			uint64_t mask = 0;
			for (size_t k=0; k<bitOf[ pi].size(); ++k) mask |= 1ull << bitOf[ pi][ k];
			T.patterns[ pi].maskLo = (uint32_t)mask; T.patterns[ pi].maskHi = (uint32_t)(mask >> 32);
			perWord[ wordOf[ pi]].push_back( (uint32_t)pi);
		}
		T.wordPatBegin.push_back( 0);
		for (uint32_t wi=0; wi<word; ++wi)
		{
			T.wordPats.insert( T.wordPats.end(), perWord[ wi].begin(), perWord[ wi].end());
			T.wordPatBegin.push_back( (uint32_t)T.wordPats.size());
		}
	}
	uint32_t nwords = word;
	T.nofPasses = (nwords + L1_WORDS_PER_PASS-1) / L1_WORDS_PER_PASS;
	if (T.nofPasses == 0) T.nofPasses = 1;
	const uint32_t totalWords = T.nofPasses * L1_WORDS_PER_PASS;
	while (T.wordPatBegin.size() < totalWords+1) T.wordPatBegin.push_back( (uint32_t)T.wordPats.size());

	T.patOfBit.assign( (size_t)totalWords*64, 0);
	for (size_t pi=0; pi<autos.size(); ++pi)
	{
		if (T.patterns[ pi].word == L1_WORD_LITERAL) continue;
		for (uint32_t k=0; k<(uint32_t)autos[ pi].pos.size(); ++k) T.patOfBit[ (size_t)T.patterns[ pi].word*64 + bitOf[ pi][ k]] = (uint32_t)pi;	// a shared bit never accepts: any owner will do
	}

	// 3. byte classes: bytes that no position distinguishes (and that share a context) are one class.  With UCP the
	//    continuation bytes 80..BF exist twice: as bytes of a character that is not a word character (symbols 128..191,
	//    context OTHER) and of one that is (symbols 256..319, context WORD) -- the kernel knows which from the lead byte.
	const unsigned nofSymbols = ucp ? 320u : 256u;
	auto symByte = []( unsigned sym) -> unsigned { return sym < 256 ? sym : 0x80u + (sym - 256u); };
	auto symCtx = []( unsigned sym) -> int { return sym < 256 ? ctxOfByte( sym) : (int)CTX_WORD; };
	// does position k of automaton a take the symbol (a byte in a context)?  Positions classed by code point take no byte.
	auto takesSymbol = [&]( const Automaton& a, size_t k, unsigned sym) -> bool
	{
		const unsigned c = symByte( sym);
		if (a.pos[ k].cpRef >= 0 || !a.pos[ k].has( c)) return false;
		return c < 0x80 || a.ctxDef[ k] < 0 || a.ctxDef[ k] == symCtx( sym);
	};
	{
		std::vector<uint64_t> sig( nofSymbols, 1469598103934665603ull);
		uint32_t gp = 0;
		for (size_t pi=0; pi<autos.size(); ++pi) for (size_t k=0; k<autos[ pi].pos.size(); ++k, ++gp)
		{
			for (unsigned c=0; c<nofSymbols; ++c)
			{
				// (a position classed by code point still tells its lead bytes apart from the others: harmless refinement)
				const bool in = autos[ pi].pos[k].cpRef >= 0 ? (c < 256 && autos[ pi].pos[k].has( c)) : takesSymbol( autos[ pi], k, c);
				if (in) sig[ c] = (sig[ c] ^ (gp+1)) * 1099511628211ull + 0x9E3779B97F4A7C15ull;
			}
		}
		std::map<std::pair<uint64_t,int>,uint32_t> classOf;
		T.byteClass.assign( nofSymbols, 0); T.classCtx.clear();
		for (unsigned c=0; c<nofSymbols; ++c)
		{
			std::pair<uint64_t,int> key( sig[ c], symCtx( c));
			std::map<std::pair<uint64_t,int>,uint32_t>::const_iterator it = classOf.find( key);
			uint32_t cls;
			if (it == classOf.end())
			{
				cls = (uint32_t)classOf.size();
				if (cls > 254) throw std::runtime_error( "too many distinct byte classes");
				classOf[ key] = cls; T.classCtx.push_back( (uint8_t)key.second);
			}
			else cls = it->second;
			T.byteClass[ c] = (uint8_t)cls;
		}
		T.nofClasses = (uint32_t)classOf.size();
	}

	// 3b. classes by code point.  A position with a code point set (ByteSet::cpRef) is entered at the lead byte of a
	//     well-formed character and looks at its code point; every other position looks at the lead byte as a byte.
	//     Code points that no position tells apart are one class, numbered behind the byte classes.
	const uint32_t nofByteClasses = T.nofClasses;
	std::vector<uint32_t> repCpOfClass;		// class id - nofByteClasses -> a code point of the class
	T.cpBlocks.clear(); T.cpPages.clear();
	T.ucp = ucp;
	if (!cpSets.empty() || ucp)
	{
		std::vector<uint32_t> cut;
		for (uint32_t cp=0x80; cp<0x800; cp+=64) cut.push_back( cp);			// one lead byte each
		for (uint32_t cp=0x800; cp<0x10000; cp=(cp+0x1000) & ~0xFFFu) cut.push_back( cp);
		for (uint32_t cp=0x10000; cp<0x110000; cp=(cp+0x40000) & ~0x3FFFFu) cut.push_back( cp);
		cut.push_back( 0x110000);
		for (size_t i=0; i<cpSets.size(); ++i) for (size_t k=0; k<cpSets[ i].size(); ++k) { cut.push_back( cpSets[ i][ k].first); cut.push_back( cpSets[ i][ k].second+1); }
		if (ucp)
		{
			// the context of a character beyond ASCII: word character or not
			static const char* cats[2] = {"L", "N"};
			for (int ci=0; ci<2; ++ci) for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
			{
				if (std::strcmp( c->name, cats[ ci])) continue;
				for (uint32_t i=0; i<c->count; ++i) { cut.push_back( c->ranges[ i].lo); cut.push_back( c->ranges[ i].hi+1); }
			}
		}
		std::sort( cut.begin(), cut.end()); cut.erase( std::unique( cut.begin(), cut.end()), cut.end());
		auto inSet = [&]( int ref, uint32_t cp) -> bool
		{
			const CpRanges& r = cpSets[ ref];
			size_t lo = 0, hi = r.size();
			while (lo < hi) { size_t mid = (lo+hi)/2; if (r[ mid].second < cp) lo = mid+1; else hi = mid; }
			return lo < r.size() && r[ lo].first <= cp;
		};
		std::vector<uint8_t> flat( 0x110000, 0xFF);
		std::map<uint64_t,uint32_t> classOfSig;
		for (size_t ai=0; ai+1<cut.size(); ++ai)
		{
			const uint32_t cp = cut[ ai];
			if (cp < 0x80 || cp >= 0x110000) continue;
			unsigned char enc[4]; encodeUtf8( cp, enc);
			const int cpCtx = (ucp && ucpWordCp( cp)) ? (int)CTX_WORD : (int)CTX_OTHER;
			uint64_t sig = 1469598103934665603ull ^ (uint64_t)cpCtx;
			uint32_t gp = 0;
			for (size_t pi=0; pi<autos.size(); ++pi) for (size_t k=0; k<autos[ pi].pos.size(); ++k, ++gp)
			{
				const ByteSet& b = autos[ pi].pos[ k];
				const bool in = (b.cpRef >= 0 ? inSet( b.cpRef, cp) : b.has( enc[0])) && (autos[ pi].ctxDef[ k] < 0 || autos[ pi].ctxDef[ k] == cpCtx);
				if (in) sig = (sig ^ (gp+1)) * 1099511628211ull + 0x9E3779B97F4A7C15ull;
			}
			std::map<uint64_t,uint32_t>::const_iterator it = classOfSig.find( sig);
			uint32_t cls;
			if (it == classOfSig.end())
			{
				cls = T.nofClasses++;
				if (cls > 254) throw std::runtime_error( "too many distinct character classes");
				classOfSig[ sig] = cls; T.classCtx.push_back( (uint8_t)cpCtx); repCpOfClass.push_back( cp);
			}
			else cls = it->second;
			for (uint32_t c=cp; c<cut[ ai+1]; ++c) flat[ c] = (uint8_t)cls;
		}
		std::map<std::string,uint16_t> pageOf;
		for (uint32_t blk=0; blk<0x110000/64; ++blk)
		{
			std::string key( (const char*)&flat[ (size_t)blk*64], 64);
			std::map<std::string,uint16_t>::const_iterator it = pageOf.find( key);
			uint16_t page;
			if (it == pageOf.end())
			{
				if (pageOf.size() >= 0xFFFF) throw std::runtime_error( "too many distinct character class pages");
				page = (uint16_t)pageOf.size(); pageOf[ key] = page;
				T.cpPages.insert( T.cpPages.end(), key.begin(), key.end());
			}
			else page = it->second;
			T.cpBlocks.push_back( page);
		}
	}

	// 4. masks
	T.charMask.assign( (size_t)T.nofPasses * T.nofClasses * 64, 0);
	T.startMask.assign( (size_t)T.nofPasses * CTX_COUNT * 64, 0);
	T.acceptMask.assign( (size_t)T.nofPasses * CTX_COUNT * 64, 0);
	T.shiftDst.assign( (size_t)T.nofPasses * 64, 0);
	T.selfLoop.assign( (size_t)T.nofPasses * 64, 0);
	std::vector<std::map<uint64_t,uint64_t> > exBySrc( totalWords);	// src bit -> dst set (edges that are neither self loop nor shift)
	std::vector<std::map<uint64_t,uint64_t> > exOfWord( totalWords);	// dst set -> src set
	std::vector<unsigned> repOfClass( T.nofClasses, 0);
	for (unsigned c=nofSymbols; c-->0;) repOfClass[ T.byteClass[ c]] = c;	// (byte classes by a symbol of theirs; the classes by code point come behind them)
	for (size_t pi=0; pi<autos.size(); ++pi)
	{
		const Automaton& a = autos[ pi];
		if (T.patterns[ pi].word == L1_WORD_LITERAL) continue;
		const uint32_t w = T.patterns[ pi].word;
		const std::vector<uint8_t>& bit = bitOf[ pi];
		const uint32_t pass = w / 64, lane = w % 64;
		const uint32_t n = (uint32_t)a.pos.size();
		auto place = [&]( uint64_t local) -> uint64_t		// pattern-local position set -> bits of the word
		{
			uint64_t rt = 0;
			for (uint32_t k=0; k<n; ++k) if (local & (1ull << k)) rt |= 1ull << bit[ k];
			return rt;
		};
		for (uint32_t k=0; k<n; ++k)
		{
			for (uint32_t cls=0; cls<T.nofClasses; ++cls)
			{
				bool member;
				if (cls < nofByteClasses) member = takesSymbol( a, k, repOfClass[ cls]);
				else
				{
					const uint32_t cp = repCpOfClass[ cls - nofByteClasses];
					if (a.pos[k].cpRef >= 0)
					{
						const CpRanges& r = cpSets[ a.pos[k].cpRef];
						member = false;
						for (size_t q=0; q<r.size() && !member; ++q) member = r[ q].first <= cp && cp <= r[ q].second;
					}
					else { unsigned char enc[4]; encodeUtf8( cp, enc); member = a.pos[k].has( enc[0]); }
					if (member && a.ctxDef[ k] >= 0 && a.ctxDef[ k] != (int)T.classCtx[ cls]) member = false;
				}
				if (member) T.charMask[ ((size_t)pass*T.nofClasses + cls)*64 + lane] |= 1ull << bit[ k];
			}
			uint64_t f = a.follow[ k];
			if (f & (1ull << k)) { T.selfLoop[ pass*64 + lane] |= 1ull << bit[ k]; f &= ~(1ull << k); }
			if (k+1 < n && (f & (1ull << (k+1))) && bit[ k+1] == bit[ k]+1) { T.shiftDst[ pass*64 + lane] |= 1ull << bit[ k+1]; f &= ~(1ull << (k+1)); }
			if (f) exBySrc[ w][ 1ull << bit[ k]] |= place( f);
		}
		for (int c=0; c<CTX_COUNT; ++c)
		{
			T.startMask[ ((size_t)pass*CTX_COUNT + c)*64 + lane] |= place( a.start[ c]);
			T.acceptMask[ ((size_t)pass*CTX_COUNT + c)*64 + lane] |= place( a.accept[ c]);
		}
	}
	// one exception row per distinct destination set of a word (sources with the same destinations share it)
	for (uint32_t w=0; w<totalWords; ++w)
	{
		for (std::map<uint64_t,uint64_t>::const_iterator it=exBySrc[ w].begin(); it!=exBySrc[ w].end(); ++it) exOfWord[ w][ it->second] |= it->first;
	}
	T.maxExceptions = 0;
	T.exCount.assign( T.nofPasses, 0);
	for (uint32_t w=0; w<totalWords; ++w)
	{
		uint32_t n = (uint32_t)exOfWord[ w].size();
		if (n > T.exCount[ w/64]) T.exCount[ w/64] = n;
		if (n > T.maxExceptions) T.maxExceptions = n;
	}
	T.exSrc.assign( (size_t)T.nofPasses * (T.maxExceptions ? T.maxExceptions : 1) * 64, 0);
	T.exDst.assign( T.exSrc.size(), 0);
	for (uint32_t w=0; w<totalWords; ++w)
	{
		uint32_t e = 0;
		for (std::map<uint64_t,uint64_t>::const_iterator it=exOfWord[ w].begin(); it!=exOfWord[ w].end(); ++it, ++e)
		{
			size_t at = ((size_t)(w/64) * (T.maxExceptions ? T.maxExceptions : 1) + e)*64 + (w%64);
			T.exDst[ at] = it->first; T.exSrc[ at] = it->second;
		}
	}

	// (the empty-match reports of ALLOWEMPTY are appended behind the automaton's reports of an offset: sort every offset's group)
	if (T.nullable.size() > 64) throw std::runtime_error( "too many expressions that match the empty string (ALLOWEMPTY: at most 64)");
	if (!T.nullable.empty()) T.reportsOrdered = false;

	// 4b. whole-word literals: hash table keyed by the word
	{
		size_t size = 1;
		while (size < literalWords.size()*2+1) size <<= 1;
		DevLiteral none; std::memset( &none, 0, sizeof(none));
		T.literals.assign( size, none);
		T.literalText.clear(); T.litPats.clear();
		T.nofLiterals = (uint32_t)literalWords.size();
		for (std::map<std::string,std::vector<uint32_t> >::const_iterator li=literalWords.begin(); li!=literalWords.end(); ++li)
		{
			uint32_t h = 0;
			for (size_t k=0; k<li->first.size(); ++k) h = h * (uint32_t)L1_LITHASH_MUL + (uint32_t)(unsigned char)li->first[k] + 1u;
			h = literalHashFinish( h);
			if (!h) h = 1;
			DevLiteral e; std::memset( &e, 0, sizeof(e));
			e.hash = h; e.textOffset = (uint32_t)T.literalText.size(); e.len = (uint32_t)li->first.size();
			e.patBegin = (uint32_t)T.litPats.size(); e.patCount = (uint32_t)li->second.size();
			T.literalText.insert( T.literalText.end(), li->first.begin(), li->first.end());
			T.litPats.insert( T.litPats.end(), li->second.begin(), li->second.end());	// definition order = ascending
			e.pat0 = li->second.front(); e.id0 = T.patterns[ e.pat0].id; e.levelBind0 = T.patterns[ e.pat0].levelBind;
			for (size_t k=0; k<li->first.size() && k<sizeof(e.text); ++k) e.text[ k] = (uint8_t)li->first[ k];
			size_t slot = h & (size-1);
			while (T.literals[ slot].hash) slot = (slot+1) & (size-1);
			T.literals[ slot] = e;
		}
		for (int pad=0; pad<4; ++pad) T.literalText.push_back( 0);	// (the kernel compares four bytes at a time)
		if (T.litPats.empty()) T.litPats.push_back( 0);
	}

	// 4c. word shapes: hash table keyed by (kind, place, literal bytes) -> expressions, ascending
	{
		std::map<std::pair<uint32_t,uint32_t>,std::vector<uint32_t> > byKey;
		for (size_t i=0; i<shapeOf.size(); ++i) byKey[ std::make_pair( shapeOf[ i].second.tag, shapeOf[ i].second.key)].push_back( shapeOf[ i].first);
		size_t size = 1;
		while (size < byKey.size()*2+1) size <<= 1;
		DevShape none; std::memset( &none, 0, sizeof(none));
		T.shapes.assign( size, none);
		T.shapePats.clear(); T.shapeVariants.clear();
		std::set<uint32_t> variants;
		for (std::map<std::pair<uint32_t,uint32_t>,std::vector<uint32_t> >::iterator ki=byKey.begin(); ki!=byKey.end(); ++ki)
		{
			std::sort( ki->second.begin(), ki->second.end());
			DevShape e; e.tag = ki->first.first; e.key = ki->first.second;
			e.patBegin = (uint32_t)T.shapePats.size(); e.patCount = (uint32_t)ki->second.size();
			T.shapePats.insert( T.shapePats.end(), ki->second.begin(), ki->second.end());
			size_t slot = shapeSlotHash( e.tag, e.key) & (size-1);
			while (T.shapes[ slot].tag) slot = (slot+1) & (size-1);
			T.shapes[ slot] = e;
			variants.insert( (e.tag & 3u) == (uint32_t)SHAPE_PREVWORD ? (uint32_t)SHAPE_PREVWORD : e.tag);
		}
		// compact form: fingerprints unique among the keys (another salt until they are)
		T.shapeSalt = 0;
		for (;; ++T.shapeSalt)
		{
			std::set<uint32_t> seen;
			bool unique = true;
			for (size_t i=0; i<T.shapes.size() && unique; ++i) if (T.shapes[ i].tag) unique = seen.insert( shapeFingerprint( T.shapes[ i].tag, T.shapes[ i].key, T.shapeSalt)).second;
			if (unique) break;
			if (T.shapeSalt > 1000) throw std::runtime_error( "internal: no salt makes the word shape fingerprints unique");
		}
		T.shapeFp.assign( T.shapes.size(), 0);
		for (size_t i=0; i<T.shapes.size(); ++i)
		{
			const DevShape& e = T.shapes[ i];
			if (!e.tag) continue;
			if (e.patCount > 255 || e.patBegin >= (1u << 24)) throw std::runtime_error( "too many expressions share one word shape key");
			const uint32_t info = (e.patCount << 24) | (e.patCount == 1 ? T.shapePats[ e.patBegin] : e.patBegin);
			T.shapeFp[ i] = (uint64_t)shapeFingerprint( e.tag, e.key, T.shapeSalt) | ((uint64_t)info << 32);
		}
		T.shapeVariants.assign( variants.begin(), variants.end());
		if (T.shapeVariants.size() > SHAPE_MAXVARIANTS) throw std::runtime_error( "internal: too many word shape variants");
		if (T.shapePats.empty()) T.shapePats.push_back( 0);
	}

	// 5. symbols: one hash table keyed by (lexem id, text)
	{
		size_t count = 0;
		for (std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator t=m_symbols.begin(); t!=m_symbols.end(); ++t) count += t->second.size();
		size_t size = 1;
		while (size < count*2+1) size <<= 1;
		DevSymbol none; std::memset( &none, 0, sizeof(none));
		T.symbols.assign( size, none);
		T.symbolText.clear();
		for (std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator t=m_symbols.begin(); t!=m_symbols.end(); ++t)
		{
			for (std::map<std::string,uint32_t>::const_iterator s=t->second.begin(); s!=t->second.end(); ++s)
			{
				uint32_t h = 2166136261u;
				for (int b=0; b<4; ++b) h = symbolHashStep( h, (t->first >> (8*b)) & 0xFF);
				for (size_t k=0; k<s->first.size(); ++k) h = symbolHashStep( h, (unsigned char)s->first[k]);
				if (!h) h = 1;
				DevSymbol e; std::memset( &e, 0, sizeof(e));
				e.hash = h; e.lexemId = t->first; e.textOffset = (uint32_t)T.symbolText.size(); e.len = (uint32_t)s->first.size(); e.symbolId = s->second;
				T.symbolText.insert( T.symbolText.end(), s->first.begin(), s->first.end());
				size_t slot = h & (size-1);
				while (T.symbols[ slot].hash) slot = (slot+1) & (size-1);
				T.symbols[ slot] = e;
			}
		}
		if (T.symbolText.empty()) T.symbolText.push_back( 0);
	}
	m_compiled = true;
}

// ---------------------------------------------------------------- compiled tables as a blob
static const char L1_MAGIC[ 9] = "SPAL1v09";

void LexCompiler::save( std::vector<uint8_t>& out) const
{
	if (!m_compiled) throw std::runtime_error( "only a compiled lexer can be serialised");
	BlobWriter w( L1_MAGIC);
	const LexTables& T = m_tables;
	w.u32( m_options);
	w.u32( T.nofPasses); w.u32( T.nofClasses); w.u32( T.maxExceptions); w.u32( T.nofLiterals); w.u32( T.nofPositions); w.u32( T.reportsOrdered ? 1u : 0u); w.u32( T.ucp ? 1u : 0u);
	w.vec( T.byteClass); w.vec( T.classCtx); w.vec( T.cpBlocks); w.vec( T.cpPages); w.vec( T.charMask); w.vec( T.startMask); w.vec( T.acceptMask); w.vec( T.shiftDst); w.vec( T.selfLoop);
	w.vec( T.exCount); w.vec( T.exSrc); w.vec( T.exDst); w.vec( T.wordPatBegin); w.vec( T.wordPats); w.vec( T.patOfBit);
	w.vec( T.patterns); w.vec( T.symbols); w.vec( T.symbolText); w.vec( T.literals); w.vec( T.literalText); w.vec( T.litPats); w.vec( T.approx); w.vec( T.nullable);
	w.u32( T.scanPasses); w.u32( T.scanWords); w.u32( T.lanesOk ? 1u : 0u); w.u32( T.nofShapes); w.vec( T.shapes); w.vec( T.shapePats); w.vec( T.shapeVariants); w.vec( T.shapeFp); w.u32( T.shapeSalt);
	w.u32( (uint32_t)m_defs.size());
	for (size_t i=0; i<m_defs.size(); ++i)
	{
		const Def& d = m_defs[ i];
		w.str( d.expression); w.u32( d.id); w.u32( d.resultIndex); w.u32( d.level); w.u32( d.editdist); w.u32( (uint32_t)d.posbind);
	}
	w.u32( (uint32_t)m_symbols.size());
	for (std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator ti=m_symbols.begin(); ti!=m_symbols.end(); ++ti)
	{
		w.u32( ti->first); w.u32( (uint32_t)ti->second.size());
		for (std::map<std::string,uint32_t>::const_iterator si=ti->second.begin(); si!=ti->second.end(); ++si) { w.str( si->first); w.u32( si->second); }
	}
	w.u32( (uint32_t)m_names.size());
	for (std::map<uint32_t,std::string>::const_iterator ni=m_names.begin(); ni!=m_names.end(); ++ni) { w.u32( ni->first); w.str( ni->second); }
	out.swap( w.finish());
}

void LexCompiler::load( const void* blob, size_t size)
{
	BlobReader r( blob, size, L1_MAGIC);
	LexTables T;
	m_options = r.u32();
	T.nofPasses = r.u32(); T.nofClasses = r.u32(); T.maxExceptions = r.u32(); T.nofLiterals = r.u32(); T.nofPositions = r.u32(); T.reportsOrdered = r.u32() != 0; T.ucp = r.u32() != 0;
	r.vec( T.byteClass); r.vec( T.classCtx); r.vec( T.cpBlocks); r.vec( T.cpPages); r.vec( T.charMask); r.vec( T.startMask); r.vec( T.acceptMask); r.vec( T.shiftDst); r.vec( T.selfLoop);
	r.vec( T.exCount); r.vec( T.exSrc); r.vec( T.exDst); r.vec( T.wordPatBegin); r.vec( T.wordPats); r.vec( T.patOfBit);
	r.vec( T.patterns); r.vec( T.symbols); r.vec( T.symbolText); r.vec( T.literals); r.vec( T.literalText); r.vec( T.litPats); r.vec( T.approx); r.vec( T.nullable);
	T.scanPasses = r.u32(); T.scanWords = r.u32(); T.lanesOk = r.u32() != 0; T.nofShapes = r.u32(); r.vec( T.shapes); r.vec( T.shapePats); r.vec( T.shapeVariants); r.vec( T.shapeFp); T.shapeSalt = r.u32();
	if (T.shapeFp.size() != T.shapes.size()) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	if (T.scanPasses > T.nofPasses || T.scanWords > T.scanPasses*64 || T.shapes.empty() || (T.shapes.size() & (T.shapes.size()-1)) || T.shapeVariants.size() > SHAPE_MAXVARIANTS || T.shapePats.empty())
	{
		throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	}
	for (size_t i=0; i<T.shapes.size(); ++i) if (T.shapes[ i].tag && (uint64_t)T.shapes[ i].patBegin + T.shapes[ i].patCount > T.shapePats.size()) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	for (size_t i=0; i<T.shapePats.size(); ++i) if (T.shapePats[ i] >= T.patterns.size() && !T.patterns.empty()) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	// the shapes the kernel indexes by must fit together (a blob of another build would fault on the device)
	if (T.byteClass.size() != (T.ucp ? 320u : 256u) || (T.ucp && T.cpBlocks.empty()) || T.classCtx.size() != T.nofClasses
	||  T.charMask.size() != (size_t)T.nofPasses*T.nofClasses*64 || T.startMask.size() != (size_t)T.nofPasses*CTX_COUNT*64 || T.acceptMask.size() != T.startMask.size()
	||  T.shiftDst.size() != (size_t)T.nofPasses*64 || T.selfLoop.size() != T.shiftDst.size() || T.exCount.size() != T.nofPasses
	||  T.exSrc.size() != (size_t)T.nofPasses*(T.maxExceptions ? T.maxExceptions : 1)*64 || T.exDst.size() != T.exSrc.size()
	||  T.patOfBit.size() != (size_t)T.nofPasses*64*64
	||  T.symbols.empty() || (T.symbols.size() & (T.symbols.size()-1)) || T.literals.empty() || (T.literals.size() & (T.literals.size()-1)))
	{
		throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	}
	for (size_t i=0; i<T.byteClass.size(); ++i) if (T.byteClass[ i] >= T.nofClasses) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	if (!T.cpBlocks.empty())
	{
		if (T.cpBlocks.size() != 0x110000/64 || T.cpPages.size() % 64) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
		for (size_t i=0; i<T.cpBlocks.size(); ++i) if ((size_t)T.cpBlocks[ i]*64 + 64 > T.cpPages.size()) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
		for (size_t i=0; i<T.cpPages.size(); ++i) if (T.cpPages[ i] != 0xFF && T.cpPages[ i] >= T.nofClasses) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	}
	m_defs.clear(); m_symbols.clear(); m_names.clear();
	const uint32_t nd = r.u32();
	for (uint32_t i=0; i<nd; ++i)
	{
		Def d; d.expression = r.str(); d.id = r.u32(); d.resultIndex = r.u32(); d.level = r.u32(); d.editdist = r.u32(); d.posbind = (int)r.u32();
		m_defs.push_back( d);
	}
	if (T.patterns.size() < m_defs.size() || (!T.approx.empty() && T.approx.size() != m_defs.size())) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	for (size_t i=0; i<T.patterns.size(); ++i)
	{
		if (T.patterns[ i].defIndex >= m_defs.size() || (i && T.patterns[ i].defIndex < T.patterns[ i-1].defIndex)) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	}
	for (size_t i=0; i<T.nullable.size(); ++i) if (T.nullable[ i].pattern >= T.patterns.size() || T.nullable.size() > 64) throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	for (size_t i=0; i<T.approx.size(); ++i)
	{
		if (T.approx[ i].len == 0 || T.approx[ i].len > L1_APPROX_MAXCHARS || T.approx[ i].editdist > L1_APPROX_MAXDIST || T.approx[ i].editdist >= T.approx[ i].len || T.approx.size() > L1_APPROX_MAXPATTERNS)
			throw std::runtime_error( "compiled lexer blob has inconsistent table shapes");
	}
	const uint32_t nt = r.u32();
	for (uint32_t i=0; i<nt; ++i)
	{
		const uint32_t id = r.u32(), n = r.u32();
		std::map<std::string,uint32_t>& tab = m_symbols[ id];
		for (uint32_t k=0; k<n; ++k) { const std::string name = r.str(); tab[ name] = r.u32(); }
	}
	const uint32_t nn = r.u32();
	for (uint32_t i=0; i<nn; ++i) { const uint32_t id = r.u32(); m_names[ id] = r.str(); }
	if (!r.atEnd()) throw std::runtime_error( "compiled lexer blob has trailing data");
	m_tables = T;
	m_compiled = true;
}
