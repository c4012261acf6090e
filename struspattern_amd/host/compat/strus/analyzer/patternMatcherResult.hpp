// Stand-ins for strusAnalyzer's PatternMatcherResult / PatternMatcherResultItem /
// PatternMatcherStatistics (strusPattern src/patternMatcher.cpp:182, :268, :307-314).
#ifndef _STRUS_ANALYZER_PATTERN_MATCHER_RESULT_HPP_INCLUDED
#define _STRUS_ANALYZER_PATTERN_MATCHER_RESULT_HPP_INCLUDED
#include "strus/analyzer/position.hpp"
#include <string>
#include <vector>
namespace strus { namespace analyzer {
class PatternMatcherResultItem
{
public:
	PatternMatcherResultItem( const char* name_, const char* value_, unsigned int ordpos_, unsigned int ordend_, const Position& origpos_, const Position& origend_)
		:m_name(name_?name_:""),m_value(value_?value_:""),m_hasvalue(value_!=0),m_ordpos(ordpos_),m_ordend(ordend_),m_origpos(origpos_),m_origend(origend_){}
	const char* name() const {return m_name.c_str();}
	const char* value() const {return m_hasvalue ? m_value.c_str() : 0;}
	unsigned int ordpos() const {return m_ordpos;}
	unsigned int ordend() const {return m_ordend;}
	const Position& origpos() const {return m_origpos;}
	const Position& origend() const {return m_origend;}
private:
	std::string m_name, m_value; bool m_hasvalue; unsigned int m_ordpos, m_ordend; Position m_origpos, m_origend;
};
class PatternMatcherResult :public PatternMatcherResultItem
{
public:
	typedef PatternMatcherResultItem Item;
	PatternMatcherResult( const char* name_, const char* value_, unsigned int ordpos_, unsigned int ordend_, const Position& origpos_, const Position& origend_, const std::vector<Item>& itemlist_=std::vector<Item>())
		:PatternMatcherResultItem(name_,value_,ordpos_,ordend_,origpos_,origend_),m_itemlist(itemlist_){}
	const std::vector<Item>& items() const {return m_itemlist;}
private:
	std::vector<Item> m_itemlist;
};
class PatternMatcherStatistics
{
public:
	class Item
	{
	public:
		Item( const char* name_, double value_) :m_name(name_),m_value(value_){}
		const char* name() const {return m_name;}
		double value() const {return m_value;}
	private:
		const char* m_name; double m_value;
	};
	void define( const char* name, double value) {m_items.push_back( Item( name, value));}
	const std::vector<Item>& items() const {return m_items;}
private:
	std::vector<Item> m_items;
};
}}
#endif
