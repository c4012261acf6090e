// TEST INFRASTRUCTURE ONLY -- CPU oracle for level 1; scope, sources and pinning: l1_oracle.hpp.
#include "l1_oracle.hpp"
#include <algorithm>
#include <cstring>
#include <limits>

using namespace oracle;

namespace oracle {

#include "unicode_categories.inc"

// ------------------------------------------------------------------ regex AST
struct Ast
{
	enum Type {Empty, Set, CpSet, Cat, Alt, Repeat, Group, Assert};
	Type type;
	uint32_t set[8];				// Set: byte set
	std::vector<std::pair<uint32_t,uint32_t> > cps;	// CpSet: code point ranges (UTF-8 mode)
	std::vector<Ast> sub;
	int rmin, rmax;					// Repeat (rmax -1 = unbounded)
	unsigned group;					// Group index
	Regex::NodeType assertion;
	Ast() :type(Empty),rmin(0),rmax(0),group(0),assertion(Regex::Eps) { std::memset( set, 0, sizeof(set)); }
};

static inline void setBit( uint32_t* s, unsigned c) { s[ c>>5] |= (1u << (c&31)); }
static inline bool hasBit( const uint32_t* s, unsigned c) { return (s[ c>>5] >> (c&31)) & 1u; }

class RegexParser
{
public:
	RegexParser( const std::string& e, unsigned options)
		:m_src(e),m_pos(0),m_opt(options),m_groups(0),m_utf8(!(options & OptByteChar)){}

	Ast parse()
	{
		Ast a = parseAlt();
		if (m_pos != m_src.size()) fail( "unbalanced ')'");
		return a;
	}
	unsigned groups() const {return m_groups;}

private:
	void fail( const char* msg) const { throw std::runtime_error( std::string("error in regular expression '") + m_src + "': " + msg); }
	bool more() const { return m_pos < m_src.size(); }
	unsigned char peek() const { return (unsigned char)m_src[ m_pos]; }

	Ast parseAlt()
	{
		Ast first = parseCat();
		if (!more() || peek() != '|') return first;
		Ast alt; alt.type = Ast::Alt; alt.sub.push_back( first);
		while (more() && peek() == '|') { ++m_pos; alt.sub.push_back( parseCat()); }
		return alt;
	}
	Ast parseCat()
	{
		Ast cat; cat.type = Ast::Cat;
		while (more() && peek() != '|' && peek() != ')') cat.sub.push_back( parseRepeat());
		if (cat.sub.empty()) return Ast();
		if (cat.sub.size() == 1) return cat.sub[0];
		return cat;
	}
	Ast parseRepeat()
	{
		Ast atom = parseAtom();
		for (;;)
		{
			if (!more()) break;
			int mn, mx;
			unsigned char c = peek();
			if (c == '*') { mn = 0; mx = -1; ++m_pos; }
			else if (c == '+') { mn = 1; mx = -1; ++m_pos; }
			else if (c == '?') { mn = 0; mx = 1; ++m_pos; }
			else if (c == '{' && parseBounds( mn, mx)) {}
			else break;
			if (more() && (peek() == '?' || peek() == '+')) ++m_pos;	// lazy / possessive: same match set
			if (atom.type == Ast::Assert) fail( "quantifier on an assertion");
			Ast rep; rep.type = Ast::Repeat; rep.rmin = mn; rep.rmax = mx; rep.sub.push_back( atom);
			atom = rep;
		}
		return atom;
	}
	bool parseBounds( int& mn, int& mx)
	{
		size_t p = m_pos+1;
		auto num = [&]( int& v) { if (p >= m_src.size() || !isdigit( (unsigned char)m_src[p])) return false; v = 0; while (p < m_src.size() && isdigit( (unsigned char)m_src[p])) { v = v*10 + (m_src[p]-'0'); if (v > 1000) fail( "repeat count too large"); ++p; } return true; };
		if (!num( mn)) return false;	// literal '{'
		mx = mn;
		if (p < m_src.size() && m_src[p] == ',')
		{
			++p;
			if (p < m_src.size() && m_src[p] == '}') mx = -1;
			else if (!num( mx)) return false;
		}
		if (p >= m_src.size() || m_src[p] != '}') return false;
		if (mx != -1 && mx < mn) fail( "bad repeat bounds");
		m_pos = p+1;
		return true;
	}
	static Ast assertion( Regex::NodeType t) { Ast a; a.type = Ast::Assert; a.assertion = t; return a; }

	// --- sets: collected as code point ranges, then lowered to bytes
	typedef std::vector<std::pair<uint32_t,uint32_t> > Ranges;
	static void addRange( Ranges& r, uint32_t lo, uint32_t hi) { r.push_back( std::make_pair( lo, hi)); }
	// UCP (PCRE): \\d = Nd, \\w = L | N | _, \\s = Z | \\h | \\v
	static void addCategory( Ranges& r, const char* name)
	{
		for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
		{
			if (std::strcmp( c->name, name)) continue;
			for (uint32_t i=0; i<c->count; ++i) addRange( r, c->ranges[i].lo, c->ranges[i].hi);
		}
	}
	void addClassEscape( Ranges& r, unsigned char e, bool& negated) const
	{
		negated = false;
		switch (e)
		{
			case 'D': negated = true; /*fall*/ case 'd': addRange( r, '0', '9'); if (m_opt & OptUcp) addCategory( r, "Nd"); break;
			case 'W': negated = true; /*fall*/ case 'w': addRange( r, '0','9'); addRange( r,'A','Z'); addRange( r,'a','z'); addRange( r,'_','_');
				if (m_opt & OptUcp) { addCategory( r, "L"); addCategory( r, "N"); }
				break;
			case 'S': negated = true; /*fall*/ case 's': addRange( r, 9, 13); addRange( r, ' ', ' ');
				if (m_opt & OptUcp) { addCategory( r, "Z"); addRange( r, 0x85, 0x85); }
				break;
		}
	}
	// \\p{Name} / \\pL / \\P{..} / \\p{^..}: a Unicode general category (m_pos is behind the 'p' or 'P')
	Ranges propertyClass( bool negated)
	{
		if (!m_utf8) fail( "unicode properties need UTF-8 mode");
		std::string name;
		if (more() && peek() == '{')
		{
			size_t e = m_src.find( '}', m_pos);
			if (e == std::string::npos) fail( "unterminated \\p{..}");
			name = m_src.substr( m_pos+1, e-m_pos-1); m_pos = e+1;
		}
		else if (more()) { name = std::string( 1, (char)peek()); ++m_pos; }
		if (!name.empty() && name[0] == '^') { negated = !negated; name.erase( 0, 1); }
		Ranges r;
		for (const UcCategory* c=UC_CATEGORIES; c->name; ++c)
		{
			if (name == c->name)
			{
				for (uint32_t i=0; i<c->count; ++i) addRange( r, c->ranges[i].lo, c->ranges[i].hi);
				return negated ? negate( r) : r;
			}
		}
		fail( "unknown unicode property");
		return r;
	}
	uint32_t maxCp() const { return m_utf8 ? 0x10FFFFu : 0xFFu; }
	static Ranges normalize( Ranges r)
	{
		std::sort( r.begin(), r.end());
		Ranges o;
		for (size_t i=0; i<r.size(); ++i)
		{
			if (!o.empty() && r[i].first <= o.back().second+1) { if (r[i].second > o.back().second) o.back().second = r[i].second; }
			else o.push_back( r[i]);
		}
		return o;
	}
	Ranges negate( const Ranges& in) const
	{
		Ranges r = normalize( in), o;
		uint32_t next = 0;
		for (size_t i=0; i<r.size(); ++i)
		{
			if (r[i].first > next) o.push_back( std::make_pair( next, r[i].first-1));
			next = r[i].second+1;
		}
		if (next <= maxCp()) o.push_back( std::make_pair( next, maxCp()));
		return o;
	}
	Ranges foldCase( const Ranges& in) const
	{
		// CASELESS: the other members of every code point's case class (unicode_categories.inc)
		if (!(m_opt & OptCaseless)) return in;
		Ranges o = in;
		for (size_t i=0; i<in.size(); ++i)
		{
			for (uint32_t k=0; k<UC_CASEPAIR_COUNT; ++k)
			{
				const UcCasePair& p = UC_CASEPAIRS[ k];
				if (p.cp < in[i].first || p.cp > in[i].second) continue;
				if (p.other > maxCp() || (!m_utf8 && p.cp > 0x7F)) continue;
				addRange( o, p.other, p.other);
			}
		}
		return o;
	}
	Ast makeSet( const Ranges& in) const
	{
		Ranges r = normalize( foldCase( in));
		Ast a;
		if (!m_utf8 || r.empty() || r.back().second < 0x80)
		{
			a.type = Ast::Set;
			for (size_t i=0; i<r.size(); ++i) for (uint32_t c=r[i].first; c<=r[i].second && c<256; ++c) setBit( a.set, c);
			return a;
		}
		a.type = Ast::CpSet; a.cps = r;
		return a;
	}
	uint32_t decodeChar()
	{
		unsigned char c = peek();
		if (!m_utf8 || c < 0x80) { ++m_pos; return c; }
		int n = (c >= 0xF0) ? 4 : (c >= 0xE0) ? 3 : (c >= 0xC0) ? 2 : 0;
		if (!n || m_pos+n > m_src.size()) fail( "invalid UTF-8 in expression");
		uint32_t cp = c & (0xFF >> (n+1));
		for (int i=1; i<n; ++i) cp = (cp << 6) | ((unsigned char)m_src[ m_pos+i] & 0x3F);
		m_pos += n;
		return cp;
	}
	uint32_t parseEscapeChar( unsigned char e)
	{
		switch (e)
		{
			case 'n': return '\n'; case 'r': return '\r'; case 't': return '\t'; case 'f': return '\f';
			case 'v': return '\v'; case 'a': return 7; case 'e': return 27; case '0': return 0;
			case 'x':
			{
				if (m_pos+2 > m_src.size()) fail( "bad \\x escape");
				unsigned v = 0;
				for (int i=0; i<2; ++i)
				{
					unsigned char h = (unsigned char)m_src[ m_pos++];
					v = v*16 + (isdigit( h) ? h-'0' : (h|32) >= 'a' && (h|32) <= 'f' ? (h|32)-'a'+10 : (fail( "bad \\x escape"), 0));
				}
				return v;
			}
			default:
				if (isalnum( e)) fail( "unsupported escape sequence");
				return e;
		}
	}
	Ast parseAtom()
	{
		unsigned char c = peek();
		switch (c)
		{
			case '(':
			{
				++m_pos;
				bool capture = true;
				if (more() && peek() == '?')
				{
					if (m_pos+1 < m_src.size() && m_src[ m_pos+1] == ':') { m_pos += 2; capture = false; }
					else fail( "unsupported group syntax");
				}
				unsigned gidx = capture ? ++m_groups : 0;
				Ast inner = parseAlt();
				if (!more() || peek() != ')') fail( "missing ')'");
				++m_pos;
				Ast g; g.type = Ast::Group; g.group = gidx; g.sub.push_back( inner);
				return g;
			}
			case '[': return parseClass();
			case '.':
			{
				++m_pos;
				Ranges r;
				if (m_opt & OptDotAll) addRange( r, 0, maxCp());
				else { addRange( r, 0, 9); addRange( r, 11, maxCp()); }
				Ast a = makeSetNoFold( r);
				return a;
			}
			case '^': ++m_pos; return assertion( (m_opt & OptMultiline) ? Regex::AssertBOL : Regex::AssertBOD);
			case '$': ++m_pos; return assertion( (m_opt & OptMultiline) ? Regex::AssertEOL : Regex::AssertEOD);
			case '\\':
			{
				++m_pos;
				if (!more()) fail( "trailing backslash");
				unsigned char e = (unsigned char)m_src[ m_pos++];
				switch (e)
				{
					case 'b': return assertion( Regex::AssertWB);
					case 'B': return assertion( Regex::AssertNWB);
					case 'A': return assertion( Regex::AssertBOD);
					case 'z': return assertion( Regex::AssertEOD);
					case 'd': case 'D': case 'w': case 'W': case 's': case 'S':
					{
						Ranges r; bool neg;
						addClassEscape( r, e, neg);
						return makeSetNoFold( neg ? negate( r) : r);
					}
					case 'p': case 'P': return makeSetNoFold( propertyClass( e == 'P'));
					default:
					{
						Ranges r; uint32_t v = parseEscapeChar( e); addRange( r, v, v);
						return makeSet( r);
					}
				}
			}
			case '*': case '+': case '?': fail( "nothing to repeat");
			case ')': fail( "unbalanced ')'");
			default:
			{
				Ranges r; uint32_t cp = decodeChar(); addRange( r, cp, cp);
				return makeSet( r);
			}
		}
		return Ast();
	}
	Ast makeSetNoFold( const Ranges& r) const
	{
		// \w \d \s and '.' are closed under ASCII case folding already
		unsigned saved = m_opt;
		const_cast<RegexParser*>(this)->m_opt &= ~(unsigned)OptCaseless;
		Ast a = makeSet( r);
		const_cast<RegexParser*>(this)->m_opt = saved;
		return a;
	}
	Ast parseClass()
	{
		++m_pos;	// '['
		bool neg = false;
		if (more() && peek() == '^') { neg = true; ++m_pos; }
		Ranges r;
		bool first = true;
		for (;;)
		{
			if (!more()) fail( "missing ']'");
			unsigned char c = peek();
			if (c == ']' && !first) { ++m_pos; break; }
			first = false;
			uint32_t lo;
			if (c == '[' && m_pos+1 < m_src.size() && m_src[ m_pos+1] == ':')
			{
				size_t e = m_src.find( ":]", m_pos+2);
				if (e == std::string::npos) fail( "bad POSIX class");
				std::string nm = m_src.substr( m_pos+2, e-m_pos-2);
				if (nm == "alpha") { addRange( r,'A','Z'); addRange( r,'a','z'); }
				else if (nm == "digit") addRange( r,'0','9');
				else if (nm == "alnum") { addRange( r,'0','9'); addRange( r,'A','Z'); addRange( r,'a','z'); }
				else if (nm == "upper") addRange( r,'A','Z');
				else if (nm == "lower") addRange( r,'a','z');
				else if (nm == "space") { addRange( r,9,13); addRange( r,' ',' '); }
				else if (nm == "punct") { addRange( r,33,47); addRange( r,58,64); addRange( r,91,96); addRange( r,123,126); }
				else if (nm == "xdigit") { addRange( r,'0','9'); addRange( r,'A','F'); addRange( r,'a','f'); }
				else fail( "unknown POSIX class");
				m_pos = e+2;
				continue;
			}
			if (c == '\\')
			{
				++m_pos;
				if (!more()) fail( "trailing backslash");
				unsigned char e = (unsigned char)m_src[ m_pos++];
				if (strchr( "dDwWsS", e))
				{
					Ranges t; bool n2;
					addClassEscape( t, e, n2);
					if (n2) t = negate( t);
					r.insert( r.end(), t.begin(), t.end());
					continue;
				}
				if (e == 'p' || e == 'P') { Ranges t = propertyClass( e == 'P'); r.insert( r.end(), t.begin(), t.end()); continue; }
				if (e == 'b') lo = 8;
				else lo = parseEscapeChar( e);
			}
			else lo = decodeChar();
			uint32_t hi = lo;
			if (m_pos+1 < m_src.size() && peek() == '-' && m_src[ m_pos+1] != ']')
			{
				++m_pos;
				if (peek() == '\\')
				{
					++m_pos;
					if (!more()) fail( "trailing backslash");
					unsigned char e = (unsigned char)m_src[ m_pos++];
					hi = parseEscapeChar( e);
				}
				else hi = decodeChar();
				if (hi < lo) fail( "bad character range");
			}
			addRange( r, lo, hi);
		}
		Ranges folded = foldCase( r);
		Ranges fin = neg ? negate( folded) : folded;
		return makeSetNoFold( fin);
	}

	std::string m_src;
	size_t m_pos;
	unsigned m_opt;
	unsigned m_groups;
	bool m_utf8;
};

// ------------------------------------------------------------------ UTF-8 lowering of code point sets
// every code point range becomes alternatives of byte-range sequences (the usual utf8-ranges split)
typedef std::vector<std::pair<unsigned char,unsigned char> > ByteSeq;
static int utf8Encode( uint32_t cp, unsigned char* b)
{
	if (cp < 0x80) { b[0] = (unsigned char)cp; return 1; }
	if (cp < 0x800) { b[0] = 0xC0 | (cp >> 6); b[1] = 0x80 | (cp & 0x3F); return 2; }
	if (cp < 0x10000) { b[0] = 0xE0 | (cp >> 12); b[1] = 0x80 | ((cp >> 6) & 0x3F); b[2] = 0x80 | (cp & 0x3F); return 3; }
	b[0] = 0xF0 | (cp >> 18); b[1] = 0x80 | ((cp >> 12) & 0x3F); b[2] = 0x80 | ((cp >> 6) & 0x3F); b[3] = 0x80 | (cp & 0x3F); return 4;
}
static void utf8Split( uint32_t lo, uint32_t hi, std::vector<ByteSeq>& out)
{
	static const uint32_t lim[3] = {0x7F, 0x7FF, 0xFFFF};
	for (int i=0; i<3; ++i)
	{
		if (lo <= lim[i] && hi > lim[i]) { utf8Split( lo, lim[i], out); utf8Split( lim[i]+1, hi, out); return; }
	}
	if (hi < 0x80) { ByteSeq s; s.push_back( std::make_pair( (unsigned char)lo, (unsigned char)hi)); out.push_back( s); return; }
	for (int i=1; i<=3; ++i)
	{
		uint32_t m = (1u << (6*i)) - 1;
		if ((lo & ~m) != (hi & ~m))
		{
			if ((lo & m) != 0) { utf8Split( lo, lo|m, out); utf8Split( (lo|m)+1, hi, out); return; }
			if ((hi & m) != m) { utf8Split( lo, (hi & ~m)-1, out); utf8Split( hi & ~m, hi, out); return; }
		}
	}
	unsigned char a[4], b[4];
	int n = utf8Encode( lo, a); utf8Encode( hi, b);
	ByteSeq s;
	for (int i=0; i<n; ++i) s.push_back( std::make_pair( a[i], b[i]));
	out.push_back( s);
}

// ------------------------------------------------------------------ Thompson construction
struct Frag { int start; std::vector<int*> outs; };

class NfaBuilder
{
public:
	explicit NfaBuilder( std::vector<Regex::Node>& nodes) :m_nodes(nodes){}
	int node( Regex::NodeType t)
	{
		Regex::Node n; n.type = t; n.out = -1; n.out1 = -1; std::memset( n.set, 0, sizeof(n.set));
		m_nodes.push_back( n);
		return (int)m_nodes.size()-1;
	}
	// fragments are described by (start node, list of dangling node ids + which out)
	struct F { int start; std::vector<std::pair<int,int> > dangling; };
	void patch( const F& f, int target)
	{
		for (size_t i=0; i<f.dangling.size(); ++i)
		{
			if (f.dangling[i].second == 0) m_nodes[ f.dangling[i].first].out = target;
			else m_nodes[ f.dangling[i].first].out1 = target;
		}
	}
	F build( const Ast& a)
	{
		switch (a.type)
		{
			case Ast::Empty: { F f; f.start = node( Regex::Eps); f.dangling.push_back( std::make_pair( f.start, 0)); return f; }
			case Ast::Set:
			{
				F f; f.start = node( Regex::Char); std::memcpy( m_nodes[ f.start].set, a.set, sizeof(a.set));
				f.dangling.push_back( std::make_pair( f.start, 0)); return f;
			}
			case Ast::CpSet:
			{
				std::vector<ByteSeq> seqs;
				for (size_t i=0; i<a.cps.size(); ++i) utf8Split( a.cps[i].first, a.cps[i].second, seqs);
				Ast alt; alt.type = Ast::Alt;
				for (size_t si=0; si<seqs.size(); ++si)
				{
					Ast cat; cat.type = Ast::Cat;
					for (size_t bi=0; bi<seqs[si].size(); ++bi)
					{
						Ast s; s.type = Ast::Set;
						for (unsigned c=seqs[si][bi].first; c<=seqs[si][bi].second; ++c) setBit( s.set, c);
						cat.sub.push_back( s);
					}
					alt.sub.push_back( cat.sub.size() == 1 ? cat.sub[0] : cat);
				}
				return build( alt.sub.size() == 1 ? alt.sub[0] : alt);
			}
			case Ast::Cat:
			{
				F f = build( a.sub[0]);
				for (size_t i=1; i<a.sub.size(); ++i) { F g = build( a.sub[i]); patch( f, g.start); f.dangling = g.dangling; }
				return f;
			}
			case Ast::Alt:
			{
				F f; f.start = -1;
				int prevSplit = -1;
				for (size_t i=0; i<a.sub.size(); ++i)
				{
					F g = build( a.sub[i]);
					f.dangling.insert( f.dangling.end(), g.dangling.begin(), g.dangling.end());
					if (i+1 < a.sub.size())
					{
						int s = node( Regex::Split); m_nodes[ s].out = g.start;
						if (prevSplit >= 0) m_nodes[ prevSplit].out1 = s; else f.start = s;
						prevSplit = s;
					}
					else
					{
						if (prevSplit >= 0) m_nodes[ prevSplit].out1 = g.start; else f.start = g.start;
					}
				}
				return f;
			}
			case Ast::Group: return build( a.sub[0]);
			case Ast::Assert: { F f; f.start = node( a.assertion); f.dangling.push_back( std::make_pair( f.start, 0)); return f; }
			case Ast::Repeat:
			{
				// x{m,n} -> x^m followed by (n-m) nested optionals, x{m,} -> x^(m-1) x+ (or x* when m==0)
				F f; bool have = false;
				auto append = [&]( const F& g) { if (!have) { f = g; have = true; } else { patch( f, g.start); f.dangling = g.dangling; } };
				int fixed = a.rmax == -1 ? (a.rmin > 0 ? a.rmin-1 : 0) : a.rmin;
				for (int i=0; i<fixed; ++i) append( build( a.sub[0]));
				if (a.rmax == -1)
				{
					F g = build( a.sub[0]);
					int s = node( Regex::Split);
					m_nodes[ s].out = g.start;
					patch( g, s);
					F loop;
					if (a.rmin > 0) loop.start = g.start; else loop.start = s;	// plus vs star
					loop.dangling.push_back( std::make_pair( s, 1));
					append( loop);
				}
				else
				{
					// optional tail: (x(x(x)?)?)?
					int opt = a.rmax - a.rmin;
					std::vector<std::pair<int,int> > exits;
					F tail; bool haveTail = false;
					for (int i=0; i<opt; ++i)
					{
						F g = build( a.sub[0]);
						int s = node( Regex::Split);
						m_nodes[ s].out = g.start;
						exits.push_back( std::make_pair( s, 1));
						if (!haveTail) { tail.start = s; haveTail = true; }
						else patch( tail, s);
						tail.dangling = g.dangling;
					}
					if (haveTail)
					{
						tail.dangling.insert( tail.dangling.end(), exits.begin(), exits.end());
						append( tail);
					}
				}
				if (!have) { f.start = node( Regex::Eps); f.dangling.push_back( std::make_pair( f.start, 0)); }
				return f;
			}
		}
		throw std::logic_error( "bad ast");
	}
private:
	std::vector<Regex::Node>& m_nodes;
};

// fixed byte length of an AST or -1
static int fixedLen( const Ast& a)
{
	switch (a.type)
	{
		case Ast::Empty: case Ast::Assert: return 0;
		case Ast::Set: return 1;
		case Ast::CpSet:
		{
			int len = -1;
			std::vector<ByteSeq> seqs;
			for (size_t i=0; i<a.cps.size(); ++i) utf8Split( a.cps[i].first, a.cps[i].second, seqs);
			for (size_t i=0; i<seqs.size(); ++i) { if (len == -1) len = (int)seqs[i].size(); else if (len != (int)seqs[i].size()) return -1; }
			return len;
		}
		case Ast::Cat: { int s = 0; for (size_t i=0; i<a.sub.size(); ++i) { int l = fixedLen( a.sub[i]); if (l < 0) return -1; s += l; } return s; }
		case Ast::Alt: { int len = -2; for (size_t i=0; i<a.sub.size(); ++i) { int l = fixedLen( a.sub[i]); if (l < 0) return -1; if (len == -2) len = l; else if (len != l) return -1; } return len; }
		case Ast::Group: return fixedLen( a.sub[0]);
		case Ast::Repeat: { if (a.rmax != a.rmin) return -1; int l = fixedLen( a.sub[0]); return l < 0 ? -1 : l*a.rmin; }
	}
	return -1;
}
// walks down to Group(g) through Cat/Group nodes only, summing fixed lengths on both sides
static bool groupContext( const Ast& a, unsigned g, int& pre, int& suf)
{
	if (a.type == Ast::Group)
	{
		if (a.group == g) return true;
		return groupContext( a.sub[0], g, pre, suf);
	}
	if (a.type == Ast::Cat)
	{
		for (size_t i=0; i<a.sub.size(); ++i)
		{
			int p = 0, s = 0;
			if (groupContext( a.sub[i], g, p, s))
			{
				for (size_t k=0; k<i; ++k) { int l = fixedLen( a.sub[k]); if (l < 0) return false; p += l; }
				for (size_t k=i+1; k<a.sub.size(); ++k) { int l = fixedLen( a.sub[k]); if (l < 0) return false; s += l; }
				pre += p; suf += s;
				return true;
			}
		}
	}
	return false;
}

} // namespace oracle

Regex::Regex( const std::string& expr, unsigned options)
{
	RegexParser parser( expr, options);
	Ast ast = parser.parse();
	m_ucp = (options & OptUcp) != 0;
	m_allowEmpty = (options & OptAllowEmpty) != 0;
	NfaBuilder b( m_nodes);
	NfaBuilder::F f = b.build( ast);
	int acc = b.node( Accept);
	b.patch( f, acc);
	m_start = f.start;
	for (unsigned g=1; g<=parser.groups(); ++g)
	{
		int pre = 0, suf = 0;
		if (groupContext( ast, g, pre, suf)) m_groupFixed[ g] = std::make_pair( pre, suf);
		else m_groupFixed[ g] = std::make_pair( -1, -1);
	}
}

bool Regex::fixedContext( unsigned group, uint32_t& prefixLen, uint32_t& suffixLen) const
{
	std::map<unsigned,std::pair<int,int> >::const_iterator it = m_groupFixed.find( group);
	if (it == m_groupFixed.end() || it->second.first < 0) return false;
	prefixLen = (uint32_t)it->second.first; suffixLen = (uint32_t)it->second.second;
	return true;
}

static inline bool isWordByte( int c) { return c >= 0 && c < 0x80 && (isalnum( c) || c == '_'); }

// Semantics of SURVEY.md App. A.2: every end offset once, leftmost start, no empty matches (with ALLOWEMPTY: also the empty
// match at an offset where nothing longer ends -- a restatement of Hyperscan's documented flag that NO vector pins).
void Regex::scan( const unsigned char* src, size_t len, std::vector<std::pair<uint32_t,uint32_t> >& out) const
{
	const uint32_t INF = 0xFFFFFFFFu;
	const size_t nn = m_nodes.size();
	std::vector<uint32_t> cur( nn, INF), nxt( nn, INF);
	std::vector<int> work;
	bool live = false;
	// UCP: \\b looks at characters: every byte of a well-formed multi-byte character is "word" iff the character is L | N
	std::vector<char> wordAt;
	if (m_ucp)
	{
		wordAt.assign( len, 0);
		for (size_t at=0; at<len;)
		{
			unsigned char c = src[ at];
			unsigned want = (c >= 0xC2 && c <= 0xDF) ? 2 : (c >= 0xE0 && c <= 0xEF) ? 3 : (c >= 0xF0 && c <= 0xF4) ? 4 : 1;
			bool ok = want > 1 && at + want <= len;
			uint32_t v = c & (0xFFu >> (want+1));
			for (unsigned k=1; ok && k<want; ++k) { if ((src[ at+k] & 0xC0) != 0x80) ok = false; else v = (v << 6) | (src[ at+k] & 0x3F); }
			if (ok) ok = want == 2 ? v >= 0x80 : want == 3 ? v >= 0x800 : (v >= 0x10000 && v <= 0x10FFFF);
			if (!ok) { wordAt[ at] = (char)(c < 0x80 && isWordByte( c)); ++at; continue; }
			bool w = false;
			static const char* cats[2] = {"L", "N"};
			for (int ci=0; ci<2 && !w; ++ci) for (const UcCategory* uc=UC_CATEGORIES; uc->name && !w; ++uc)
			{
				if (std::strcmp( uc->name, cats[ ci])) continue;
				for (uint32_t k=0; k<uc->count && !w; ++k) w = uc->ranges[k].lo <= v && v <= uc->ranges[k].hi;
			}
			for (unsigned k=0; k<want; ++k) wordAt[ at+k] = (char)w;
			at += want;
		}
	}
	auto isWordPrev = [&]( size_t i) -> bool { return i > 0 && (m_ucp ? wordAt[ i-1] != 0 : isWordByte( src[ i-1])); };
	auto isWordNext = [&]( size_t i) -> bool { return i < len && (m_ucp ? wordAt[ i] != 0 : isWordByte( src[ i])); };
	for (size_t i=0; i<=len; ++i)
	{
		int prev = i > 0 ? src[ i-1] : -1;
		int next = i < len ? src[ i] : -1;
		// inject a new thread at every offset (unanchored search) and close over epsilon edges
		if (cur[ m_start] > (uint32_t)i) cur[ m_start] = (uint32_t)i;
		work.clear();
		for (size_t n=0; n<nn; ++n) if (cur[ n] != INF && m_nodes[ n].type != Char) work.push_back( (int)n);
		(void)live;
		while (!work.empty())
		{
			int n = work.back(); work.pop_back();
			const Node& nd = m_nodes[ n];
			uint32_t s = cur[ n];
			bool pass = true;
			switch (nd.type)
			{
				case Char: case Accept: continue;
				case Eps: case Split: break;
				case AssertWB: pass = (isWordPrev( i) != isWordNext( i)); break;
				case AssertNWB: pass = (isWordPrev( i) == isWordNext( i)); break;
				case AssertBOL: pass = (prev == -1 || prev == '\n'); break;
				case AssertEOL: pass = (next == -1 || next == '\n'); break;
				case AssertBOD: pass = (prev == -1); break;
				case AssertEOD: pass = (next == -1); break;
			}
			if (!pass) continue;
			int targets[2] = { nd.out, nd.type == Split ? nd.out1 : -1 };
			for (int t=0; t<2; ++t)
			{
				int o = targets[t];
				if (o >= 0 && s < cur[ o]) { cur[ o] = s; if (m_nodes[ o].type != Char) work.push_back( o); }
			}
		}
		// report
		for (size_t n=0; n<nn; ++n)
		{
			// (HS_FLAG_ALLOWEMPTY: also the empty match that starts and ends here, when nothing longer ends here)
			if (m_nodes[ n].type == Accept && cur[ n] != INF && (cur[ n] < (uint32_t)i || m_allowEmpty)) out.push_back( std::make_pair( cur[ n], (uint32_t)i));
		}
		if (i == len) break;
		// consume src[i]
		std::fill( nxt.begin(), nxt.end(), INF);
		unsigned c = src[ i];
		for (size_t n=0; n<nn; ++n)
		{
			if (cur[ n] != INF && m_nodes[ n].type == Char && hasBit( m_nodes[ n].set, c))
			{
				int o = m_nodes[ n].out;
				if (cur[ n] < nxt[ o]) nxt[ o] = cur[ n];
			}
		}
		cur.swap( nxt);
	}
}

// ------------------------------------------------------------------ LexerInstance
static bool ieq1( const std::string& a, const char* b)
{
	size_t n = std::strlen( b);
	if (a.size() != n) return false;
	for (size_t i=0; i<n; ++i) if (toupper( (unsigned char)a[i]) != toupper( (unsigned char)b[i])) return false;
	return true;
}

// patternLexer.cpp:605-626
static unsigned extractEditDist( std::string& expr)
{
	if (expr.empty()) return 0;
	const char* si = expr.c_str();
	const char* se = si + expr.size();
	unsigned dcnt = 0;
	for (--se; se >= si && (unsigned char)*se <= 32; --se){}
	for (; se >= si && *se >= '0' && *se <= '9'; --se,++dcnt){}
	if (dcnt > 0)
	{
		const char* ediststr = se+1;
		for (; se >= si && (unsigned char)*se <= 32; --se){}
		if (se >= si && *se == '~')
		{
			unsigned rt = (unsigned)atoi( ediststr);
			for (; se > si && (unsigned char)*(se-1) <= 32; --se){}
			expr.resize( se-si);
			return rt;
		}
	}
	return 0;
}

// patternLexer.cpp:245-262, :990-1006
void LexerInstance::defineLexem( uint32_t id, const std::string& expression, uint32_t resultIndex, uint32_t level, PosBind posbind)
{
	if (m_compiled) throw std::runtime_error( "called define pattern after calling 'compile'");
	if (id > (1u<<30)-1) throw std::runtime_error( "pattern id out of range");
	if (level > 255 || resultIndex > 255) throw std::runtime_error( "level or result index out of range");
	Def d; d.expression = expression; d.editdist = extractEditDist( d.expression);
	d.id = id; d.resultIndex = resultIndex; d.level = level; d.posbind = posbind; d.prefixLen = 0; d.suffixLen = 0;
	m_defs.push_back( d);
}

// patternLexer.cpp:264-293
void LexerInstance::defineSymbol( uint32_t symbolid, uint32_t patternid, const std::string& name)
{
	if (m_compiled) throw std::runtime_error( "called define pattern after calling 'compile'");
	std::map<std::string,uint32_t>& tab = m_symbols[ patternid];
	if (tab.count( name)) throw std::runtime_error( "symbol defined twice: '" + name + "'");
	tab[ name] = symbolid;
}

// patternLexer.cpp:295-310
uint32_t LexerInstance::getSymbol( uint32_t patternid, const std::string& name) const
{
	std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator ti = m_symbols.find( patternid);
	if (ti == m_symbols.end()) return 0;
	std::map<std::string,uint32_t>::const_iterator si = ti->second.find( name);
	return si == ti->second.end() ? 0 : si->second;
}

// patternLexer.cpp:1031-1066
void LexerInstance::defineOption( const std::string& name, double)
{
	if (ieq1( name, "CASELESS")) m_options |= OptCaseless;
	else if (ieq1( name, "DOTALL")) m_options |= OptDotAll;
	else if (ieq1( name, "MULTILINE")) m_options |= OptMultiline;
	else if (ieq1( name, "ALLOWEMPTY")) m_options |= OptAllowEmpty;
	else if (ieq1( name, "UCP")) m_options |= OptUcp;
	else if (ieq1( name, "BYTECHAR")) m_options |= OptByteChar;
	else throw std::runtime_error( "unknown option '" + name + "'");
}

// ------------------------------------------------------------------ approximate literal tables
// A table with at least one `~N` expression takes the reference's other route through its two libraries
// (patternLexer.cpp:333-412): EVERY expression is pre-matched by Hyperscan on the one-byte-per-character
// hash of the text (OneByteCharMap, unicodeUtils.cpp:19-44: a character <= 127 is itself, any other is
// 128 + code point % 128) and every candidate is re-matched by libtre on wide characters
// (SubExpressionDef, :450-601).  Restated for expressions that are plain literals:
//   stage 1 (Hyperscan, approximate matching with leftmost start of match): for every pattern and every
//     character end position e, a candidate (s,e) exists when some s < e has
//     levenshtein( hash(text[s:e]), hash(literal)) <= N; s is the smallest such start (N = 0: the
//     hashed literal itself ends at e).  Candidates are delivered by end position, then pattern index.
//   stage 2 (libtre): an edit distance pattern is searched approximately (unit costs, maximum cost
//     N+3, :527-535) in the window of the candidate's bytes plus N*4 further bytes (:561); an exact
//     pattern is searched exactly from the candidate's start to the end of the document and must end
//     inside the candidate (:505).  The match found replaces the candidate's (from,to).
// MODEL (what libtre returns among several approximate matches is not derivable from the reference's
// sources): the window is read left to right with the usual states (pattern characters consumed, cost,
// start); pattern characters may be skipped (cost 1) before a character is read; a match is recorded
// when the last pattern character is consumed by reading a character (exactly or as a substitution), or
// at the end of the window; a later match replaces the recorded one only if it is strictly cheaper.
// This reproduces both vectors of testCharRegexMatch.cpp:161-196, which is all that pins it.
static bool decodeChar( const unsigned char* s, size_t len, size_t at, uint32_t& cp, unsigned& n)
{
	// lenient UTF-8: a lead byte with all its continuation bytes is one character, any other byte is a
	// character of its own value
	unsigned char c = s[ at];
	n = 1; cp = c;
	unsigned want = (c >= 0xC2 && c <= 0xDF) ? 2 : (c >= 0xE0 && c <= 0xEF) ? 3 : (c >= 0xF0 && c <= 0xF4) ? 4 : 1;
	if (want == 1 || at + want > len) return true;
	uint32_t v = c & (0xFFu >> (want+1));
	for (unsigned i=1; i<want; ++i)
	{
		if ((s[ at+i] & 0xC0) != 0x80) return true;
		v = (v << 6) | (s[ at+i] & 0x3F);
	}
	cp = v; n = want;
	return true;
}
static inline uint32_t oneByteHash( uint32_t cp) { return cp <= 127 ? cp : 128 + (cp % 128); }

static unsigned levenshtein( const uint32_t* a, size_t na, const uint32_t* b, size_t nb)
{
	std::vector<unsigned> prev( nb+1), cur( nb+1);
	for (size_t j=0; j<=nb; ++j) prev[ j] = (unsigned)j;
	for (size_t i=1; i<=na; ++i)
	{
		cur[ 0] = (unsigned)i;
		for (size_t j=1; j<=nb; ++j)
		{
			unsigned v = prev[ j-1] + (a[ i-1] != b[ j-1] ? 1u : 0u);
			if (prev[ j] + 1 < v) v = prev[ j] + 1;
			if (cur[ j-1] + 1 < v) v = cur[ j-1] + 1;
			cur[ j] = v;
		}
		prev.swap( cur);
	}
	return prev[ nb];
}

// the second stage's approximate search: (start, end) in characters of `w`, false if nothing within maxCost
static bool approxSearch( const std::vector<uint32_t>& w, const std::vector<uint32_t>& lit, unsigned maxCost, size_t& ms, size_t& me)
{
	const size_t m = lit.size();
	const unsigned NONE = 0xFFFFFFFFu;
	std::vector<unsigned> cost( m+1, NONE), ncost( m+1, NONE);
	std::vector<size_t> start( m+1, 0), nstart( m+1, 0);
	bool have = false; unsigned best = NONE;
	size_t pos = 0;
	for (;;)
	{
		if (!have || best > 0)
		{
			if (cost[ 0] == NONE || cost[ 0] > 0) { cost[ 0] = 0; start[ 0] = pos; }
		}
		// pattern characters skipped
		for (size_t k=0; k<m; ++k)
		{
			if (cost[ k] == NONE) continue;
			unsigned c = cost[ k] + 1;
			if (c > maxCost || (have && c >= best)) continue;
			if (cost[ k+1] == NONE || c < cost[ k+1]) { cost[ k+1] = c; start[ k+1] = start[ k]; }
		}
		if (pos == w.size())
		{
			if (cost[ m] != NONE && (!have || cost[ m] < best)) { have = true; best = cost[ m]; ms = start[ m]; me = pos; }
			break;
		}
		const uint32_t ch = w[ pos]; ++pos;
		std::fill( ncost.begin(), ncost.end(), NONE);
		for (size_t k=0; k<m; ++k)			// (the final state has no transitions)
		{
			if (cost[ k] == NONE) continue;
			// the character is the next pattern character, or stands for it
			{
				unsigned c = cost[ k] + (ch != lit[ k] ? 1u : 0u);
				if (c <= maxCost && !(have && c >= best))
				{
					if (ncost[ k+1] == NONE || c < ncost[ k+1]) { ncost[ k+1] = c; nstart[ k+1] = start[ k]; }
					if (k+1 == m && (!have || c < best)) { have = true; best = c; ms = start[ k]; me = pos; }
				}
			}
			// the character is an extra one
			{
				unsigned c = cost[ k] + 1;
				if (c <= maxCost && !(have && c >= best))
				{
					if (ncost[ k] == NONE || c < ncost[ k]) { ncost[ k] = c; nstart[ k] = start[ k]; }
				}
			}
		}
		cost.swap( ncost); start.swap( nstart);
	}
	return have;
}

bool LexerInstance::approxSecondStage( const char* src, size_t len, const ApproxCandidate& c, uint32_t& from, uint32_t& to) const
{
	const Def& def = m_defs[ c.idx-1];
	const std::vector<uint32_t>& lit = m_literal[ c.idx-1];
	const unsigned char* s = (const unsigned char*)src;
	if (def.editdist)
	{
		// WCharString( src+from, to-from + editdist*sizeof(wchar_t)) (:561): whole characters of that many bytes
		size_t wend = (size_t)c.to + 4u * def.editdist;
		if (wend > len) wend = len;
		std::vector<uint32_t> w; std::vector<size_t> off;
		for (size_t at=c.from; at<wend;)
		{
			uint32_t cp; unsigned n; decodeChar( s, wend, at, cp, n);
			off.push_back( at); w.push_back( cp); at += n;
		}
		off.push_back( wend);
		size_t ms, me;
		if (!approxSearch( w, lit, def.editdist + 3, ms, me)) return false;
		from = (uint32_t)off[ ms]; to = (uint32_t)off[ me];
		return true;
	}
	// exact re-match: leftmost occurrence from the candidate's start on, must end inside the candidate (:505)
	std::string bytes;
	for (size_t k=0; k<lit.size(); ++k)
	{
		unsigned char b[ 4]; int n = utf8Encode( lit[ k], b);
		bytes.append( (const char*)b, n);
	}
	for (size_t at=c.from; at + bytes.size() <= len; ++at)
	{
		if (std::memcmp( s + at, bytes.data(), bytes.size()) == 0)
		{
			if (at + bytes.size() > c.to) return false;
			from = (uint32_t)at; to = (uint32_t)(at + bytes.size());
			return true;
		}
	}
	return false;
}

std::vector<LexemOut> LexerInstance::matchApprox( const char* src, size_t len) const
{
	const unsigned char* s = (const unsigned char*)src;
	// OneByteCharMap::init: hashed text and the byte offset of every character
	std::vector<uint32_t> hashed; std::vector<uint32_t> posar;
	for (size_t at=0; at<len;)
	{
		uint32_t cp; unsigned n; decodeChar( s, len, at, cp, n);
		posar.push_back( (uint32_t)at); hashed.push_back( oneByteHash( cp)); at += n;
	}
	posar.push_back( (uint32_t)len);
	std::vector<ApproxCandidate> cand;
	for (size_t e=1; e<=hashed.size(); ++e)
	{
		for (size_t pi=0; pi<m_defs.size(); ++pi)
		{
			std::vector<uint32_t> hl;
			for (size_t k=0; k<m_literal[ pi].size(); ++k) hl.push_back( oneByteHash( m_literal[ pi][ k]));
			const size_t m = hl.size(), N = m_defs[ pi].editdist;
			for (size_t L = m+N < e ? m+N : e; L >= 1 && L + N >= m; --L)
			{
				if (levenshtein( &hashed[ e-L], L, hl.data(), m) <= N)
				{
					ApproxCandidate c; c.idx = (uint32_t)pi+1; c.from = posar[ e-L]; c.to = posar[ e];
					cand.push_back( c);
					break;
				}
			}
		}
	}
	std::vector<MatchEvent> ar;
	for (size_t i=0; i<cand.size(); ++i)
	{
		if (cand[i].to - cand[i].from >= 65535) throw std::runtime_error( "size of matched term out of range");
		uint32_t from = 0, to = 0;
		if (!approxSecondStage( src, len, cand[i], from, to)) continue;
		handleEvent( ar, src, cand[i].idx, from, to);
	}
	return ordinalPositions( ar);
}

static bool plainLiteral( const std::string& expr)
{
	if (expr.empty()) return false;
	for (size_t i=0; i<expr.size(); ++i) if (std::strchr( "\\.[](){}|*+?^$", expr[i])) return false;
	return true;
}

// patternLexer.cpp:1068-1118 + :333-412.  BYTECHAR / UCP / ALLOWEMPTY belong to the "next" rows of
// SURVEY.md 8(f) and are rejected here.
void LexerInstance::compile()
{
	m_regex.clear(); m_literal.clear(); m_approx = false;
	for (size_t i=0; i<m_defs.size(); ++i) if (m_defs[i].editdist) m_approx = true;
	if (m_options & OptByteChar) m_approx = true;		// forceOneByteCharMap (patternLexer.cpp:1055-1058): the same route
	if (m_approx)
	{
		if (m_options & OptCaseless) throw std::runtime_error( "edit distance matching (~N) with CASELESS is not supported by this oracle");
		if (!m_symbols.empty()) throw std::runtime_error( "edit distance matching (~N) with symbols is not supported by this oracle");
		for (size_t i=0; i<m_defs.size(); ++i)
		{
			const Def& d = m_defs[i];
			if (!plainLiteral( d.expression) || d.resultIndex) throw std::runtime_error( "a table with edit distance matching (~N) holds plain literal expressions only: " + d.expression);
			std::vector<uint32_t> cps;
			const unsigned char* s = (const unsigned char*)d.expression.data();
			for (size_t at=0; at<d.expression.size();) { uint32_t cp; unsigned n; decodeChar( s, d.expression.size(), at, cp, n); cps.push_back( cp); at += n; }
			if (cps.size() > 24 || d.editdist > 3 || (d.editdist && d.editdist >= cps.size())) throw std::runtime_error( "edit distance literal: at most 24 characters, distance at most 3 and below the length: " + d.expression);
			m_literal.push_back( cps);
		}
		m_compiled = true;
		return;
	}
	for (size_t i=0; i<m_defs.size(); ++i)
	{
		m_regex.push_back( Regex( m_defs[i].expression, m_options));
		if (!(m_options & OptAllowEmpty))
		{
			// Hyperscan refuses an expression that matches the empty buffer unless HS_FLAG_ALLOWEMPTY is given
			Regex probe( m_defs[i].expression, m_options | OptAllowEmpty);
			std::vector<std::pair<uint32_t,uint32_t> > hit;
			probe.scan( (const unsigned char*)"", 0, hit);
			if (!hit.empty()) throw std::runtime_error( "Pattern matches empty buffer; use option ALLOWEMPTY to enable support: " + m_defs[i].expression);
		}
		if (m_defs[i].resultIndex)
		{
			// sub-expression selection (patternLexer.cpp:488-507 via libtre): supported when the text
			// around the selected group has a fixed length, then from += prefix, to -= suffix
			if (!m_regex.back().fixedContext( m_defs[i].resultIndex, m_defs[i].prefixLen, m_defs[i].suffixLen))
				throw std::runtime_error( "sub-expression selection needs fixed-length context around the group: " + m_defs[i].expression);
		}
	}
	m_compiled = true;
}

std::vector<RawMatch> LexerInstance::rawMatches( const char* src, size_t len) const
{
	std::vector<RawMatch> raw;
	std::vector<std::pair<uint32_t,uint32_t> > one;
	for (size_t pi=0; pi<m_regex.size(); ++pi)
	{
		one.clear();
		m_regex[ pi].scan( (const unsigned char*)src, len, one);
		for (size_t k=0; k<one.size(); ++k) { RawMatch r; r.idx = (uint32_t)pi+1; r.from = one[k].first; r.to = one[k].second; raw.push_back( r); }
	}
	// Hyperscan delivers reports in order of the end offset; ties in ascending pattern index
	// (pinned by testCharRegexMatch.cpp:155-156)
	std::stable_sort( raw.begin(), raw.end(), []( const RawMatch& a, const RawMatch& b){ return a.to != b.to ? a.to < b.to : a.idx < b.idx; });
	return raw;
}

// patternLexer.cpp:717-826, literal restatement on a std::vector (A.3)
void LexerInstance::handleMatch( std::vector<MatchEvent>& ar, const char* src, uint32_t idx, uint32_t from, uint32_t to) const
{
	if (to - from >= 65535) throw std::runtime_error( "size of matched term out of range");
	const Def& def = m_defs[ idx-1];
	if (def.resultIndex)
	{
		if (def.prefixLen + def.suffixLen > to - from) return;
		from += def.prefixLen; to -= def.suffixLen;
	}
	handleEvent( ar, src, idx, from, to);
}

// the handler behind the sub-expression step (:739-826)
void LexerInstance::handleEvent( std::vector<MatchEvent>& ar, const char* src, uint32_t idx, uint32_t from, uint32_t to) const
{
	const Def& def = m_defs[ idx-1];
	uint32_t patternid = def.id;
	std::map<uint32_t, std::map<std::string,uint32_t> >::const_iterator ti = m_symbols.find( def.id);
	if (ti != m_symbols.end())
	{
		std::map<std::string,uint32_t>::const_iterator si = ti->second.find( std::string( src+from, to-from));
		if (si != ti->second.end() && si->second) patternid = si->second;
	}
	MatchEvent ev; ev.id = def.id; ev.level = (uint8_t)def.level; ev.posbind = (uint8_t)def.posbind;
	ev.origpos = from; ev.origsize = (uint16_t)(to-from);
	MatchEvent twin = ev; twin.id = patternid;
	if (ar.empty())
	{
		ar.push_back( ev);
		if (patternid != def.id) ar.push_back( twin);
		return;
	}
	size_t nofDeletes = 0;
	uint32_t matchLastPos = ev.origpos + ev.origsize;
	// delete pass: from the back while origpos >= new origpos (:759-777).  The reference shifts the
	// tail down in place and shrinks the vector afterwards; deleting in place while walking
	// backwards visits the same elements.
	for (size_t k=ar.size(); k>0; --k)
	{
		const MatchEvent& m = ar[ k-1];
		if (!(m.origpos >= ev.origpos)) break;
		if ((ev.id == m.id && m.origpos == ev.origpos && m.level == ev.level)
		||  (m.level < ev.level && m.origpos + m.origsize <= matchLastPos))
		{
			ar.erase( ar.begin() + (k-1));
			++nofDeletes;
		}
	}
	if (!nofDeletes)
	{
		// ignore pass (:778-792)
		for (size_t k=ar.size(); k>0; --k)
		{
			const MatchEvent& m = ar[ k-1];
			if (!(m.origpos + m.origsize >= matchLastPos)) break;
			if (m.level > ev.level && m.origpos <= ev.origpos) return;
		}
	}
	// insert in ascending order of origpos (:793-822) -- literal restatement of the reverse-iterator
	// shifting loops.  NOTE: the two-element (symbol twin) variant of the reference shifts the first
	// visited element by two slots but every further one by one slot only, so inserting a lexem+symbol
	// pair two or more places before the end drops an event and leaves a zero event behind.  That is
	// what the reference computes, so it is what is restated here.
	const size_t n = ar.size();
	MatchEvent zero; std::memset( &zero, 0, sizeof(zero));
	if (patternid == def.id)
	{
		ar.resize( n+1, zero);
		long prev = (long)n, mi = (long)n-1;
		for (; mi >= 0 && ar[ mi].origpos > ev.origpos; prev = mi--) ar[ prev] = ar[ mi];
		++mi;
		ar[ mi] = ev;
	}
	else
	{
		ar.resize( n+2, zero);
		long prev = (long)n+1, mi = (long)n-1;
		for (; mi >= 0 && ar[ mi].origpos > ev.origpos; prev = mi--) ar[ prev] = ar[ mi];
		++mi;
		ar[ mi] = ev;
		++mi;
		ar[ mi] = twin;
	}
}

// patternLexer.cpp:858-950
std::vector<LexemOut> LexerInstance::match( const char* src, size_t len) const
{
	if (!m_compiled) throw std::runtime_error( "called create context without calling 'compile'");
	if (len >= 0xFFFFFFFFull) throw std::runtime_error( "size of string to scan out of range");
	if (m_approx) return matchApprox( src, len);
	std::vector<RawMatch> raw = rawMatches( src, len);
	std::vector<MatchEvent> ar;
	for (size_t i=0; i<raw.size(); ++i) handleMatch( ar, src, raw[i].idx, raw[i].from, raw[i].to);
	return ordinalPositions( ar);
}

// ordinal positions (:893-945)
std::vector<LexemOut> LexerInstance::ordinalPositions( const std::vector<MatchEvent>& ar) const
{
	std::vector<LexemOut> rt;
	size_t mi = 0;
	uint32_t ordpos = 0, origpos = 0;
	uint8_t lastposbind = BindContent;
	for (; mi < ar.size(); ++mi)
	{
		lastposbind = ar[mi].posbind;
		if (ar[mi].posbind == BindUnique || ar[mi].posbind == BindContent)
		{
			ordpos = 1; origpos = ar[mi].origpos;
			LexemOut l = { ar[mi].id, 1, ar[mi].origpos, ar[mi].origsize }; rt.push_back( l);
			++mi;
			break;
		}
		else if (ar[mi].posbind == BindSuccessor)
		{
			LexemOut l = { ar[mi].id, 1, ar[mi].origpos, ar[mi].origsize }; rt.push_back( l);
		}
	}
	if (ordpos == 0) rt.clear();
	for (; mi < ar.size(); ++mi)
	{
		const MatchEvent& m = ar[mi];
		switch (m.posbind)
		{
			case BindUnique:
				if (lastposbind == BindUnique) break;
				/* fall through */
			case BindContent:
				if (m.origpos > origpos) { origpos = m.origpos; ++ordpos; }
				{ LexemOut l = { m.id, ordpos, m.origpos, m.origsize }; rt.push_back( l); }
				break;
			case BindSuccessor:
				{ LexemOut l = { m.id, ordpos+1, m.origpos, m.origsize }; rt.push_back( l); }
				break;
			case BindPredecessor:
				{ LexemOut l = { m.id, ordpos, m.origpos, m.origsize }; rt.push_back( l); }
				break;
		}
		lastposbind = m.posbind;
	}
	return rt;
}
