// Stand-in for strusAnalyzer's analyzer::Position.
#ifndef _STRUS_ANALYZER_POSITION_HPP_INCLUDED
#define _STRUS_ANALYZER_POSITION_HPP_INCLUDED
namespace strus { namespace analyzer {
class Position
{
public:
	Position() :m_seg(0),m_ofs(0){}
	Position( int seg_, int ofs_) :m_seg(seg_),m_ofs(ofs_){}
	int seg() const {return m_seg;}
	int ofs() const {return m_ofs;}
private:
	int m_seg; int m_ofs;
};
enum PositionBind {BindContent, BindSuccessor, BindPredecessor, BindUnique};
}}
#endif
