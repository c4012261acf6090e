// Stand-in for strusBase's StructView (introspection), only the ("key",value) builder form used
// by strusPattern (src/patternLexer.cpp:1140, :1178-1180).
#ifndef _STRUS_STRUCT_VIEW_HPP_INCLUDED
#define _STRUS_STRUCT_VIEW_HPP_INCLUDED
#include <map>
#include <string>
namespace strus {
class StructView
{
public:
	StructView(){}
	StructView( int){}
	StructView& operator()( const char* key, const char* value) {m_map[ key] = value ? value : ""; return *this;}
	const std::map<std::string,std::string>& dict() const {return m_map;}
private:
	std::map<std::string,std::string> m_map;
};
}
#endif
