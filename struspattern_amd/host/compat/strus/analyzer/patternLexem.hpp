// Stand-in for strusAnalyzer's analyzer::PatternLexem (strusPattern src/patternLexer.cpp:908).
#ifndef _STRUS_ANALYZER_PATTERN_LEXEM_HPP_INCLUDED
#define _STRUS_ANALYZER_PATTERN_LEXEM_HPP_INCLUDED
#include "strus/analyzer/position.hpp"
#include <cstddef>
namespace strus { namespace analyzer {
class PatternLexem
{
public:
	PatternLexem( unsigned int id_, unsigned int ordpos_, const Position& origpos_, std::size_t origsize_)
		:m_id(id_),m_ordpos(ordpos_),m_origpos(origpos_),m_origsize(origsize_){}
	unsigned int id() const {return m_id;}
	unsigned int ordpos() const {return m_ordpos;}
	const Position& origpos() const {return m_origpos;}
	std::size_t origsize() const {return m_origsize;}
private:
	unsigned int m_id; unsigned int m_ordpos; Position m_origpos; std::size_t m_origsize;
};
}}
#endif
