"""Compiled-table serialisation: a lexer / matcher saved to a blob and loaded again has the
same tables word for word, keeps names, symbols, format strings and options, and rejects damaged blobs."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth


def _matcher():
    m = spa.PatternMatcherInstance()
    m.defineOption("maxResultSize", 77)
    m.defineOption("exclusive")
    rules = synth.random_rules(400, 60, 5)
    synth.apply_rules(m, rules, compile=False)
    m.pushTerm(3)
    m.attachVariable("x")
    m.pushTerm(4)
    m.pushExpression("sequence", 2, 4, 0)
    m.definePattern("with_format", "{x} then four", True)
    m.compile()
    return m


def test_matcher_round_trip():
    m = _matcher()
    blob = m.serialize()
    n = spa.PatternMatcherInstance.deserialize(blob)
    assert np.array_equal(m.dumpTable(), n.dumpTable())
    assert n.patternId("with_format") == m.patternId("with_format") and n.patternName(m.patternId("sequence_0")) == "sequence_0"
    assert n.variableName(m.variableId("x")) == "x"
    assert n.formatCount() == 1 and n.formatString(1) == "{x} then four"
    assert n.fastTier() == m.fastTier()
    assert n.serialize() == blob                     # a loaded rule set saves to the same bytes


def test_lexer_round_trip():
    vocab = synth.vocabulary(500, 3)
    pats = synth.lexer_patterns(300, vocab, 3)
    lx = spa.PatternLexerInstance()
    for o in ("DOTALL",):
        lx.defineOption(o)
    for lid, expr, residx, level, posbind in pats:
        lx.defineLexem(lid, expr, residx, level, posbind)
    lx.defineLexemName(7, "seven")
    lx.defineSymbol(9001, 7, vocab[6])
    with pytest.raises(spa.PatternError):
        lx.serialize()                               # only a compiled lexer can be saved
    lx.compile()
    blob = lx.serialize()
    ly = spa.PatternLexerInstance.deserialize(blob)
    assert np.array_equal(lx.dumpTables(), ly.dumpTables())
    assert ly.getLexemName(7) == "seven" and ly.getSymbol(7, vocab[6]) == 9001
    assert ly.serialize() == blob
    with pytest.raises(spa.PatternError):
        ly.defineLexem(1, "x")                       # compiled and frozen, as after compile()


def test_damaged_blobs_are_rejected():
    blob = _matcher().serialize()
    for bad in (blob[:-9], blob[:40], b"", b"garbage" * 10, blob[:100] + bytes([blob[100] ^ 1]) + blob[101:]):
        with pytest.raises(spa.PatternError):
            spa.PatternMatcherInstance.deserialize(bad)
    with pytest.raises(spa.PatternError):
        spa.PatternLexerInstance.deserialize(blob)   # a rule set is not a lexer


@pytest.mark.gpu
def test_loaded_tables_match_like_the_compiled_ones():
    vocab = synth.vocabulary(2000, 5)
    pats, rules = synth.pipeline_workload(300, 600, vocab, 6)
    text, offs = synth.text_documents(12, 3000, vocab, 7, utf8=True)
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    m = spa.PatternMatcherInstance()
    synth.apply_rules(m, rules)
    lx2 = spa.PatternLexerInstance.deserialize(lx.serialize())
    m2 = spa.PatternMatcherInstance.deserialize(m.serialize())
    a = lx.createContext().matchDocs(text, offs)
    b = lx2.createContext().matchDocs(text, offs)
    assert np.array_equal(a.lexems, b.lexems) and np.array_equal(a.doc_offsets, b.doc_offsets)
    ra = m.createContext().matchDocs(a.lexems, a.doc_offsets)
    rb = m2.createContext().matchDocs(b.lexems, b.doc_offsets)
    assert len(ra.results) > 0
    assert np.array_equal(ra.results, rb.results) and np.array_equal(ra.items, rb.items) and np.array_equal(ra.stats, rb.stats)
