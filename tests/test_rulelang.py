"""The rule-language loader (struspattern_amd/rulelang.py) on the worked example of the reference's
web page (tests/golden/doc_example.json): program, input text, the published token listing of
`strusPatternMatch -K` and the published result listing.

CPU tests: the oracle behind the loader must reproduce both listings literally (this pins the lexer
restatement -- levels / supersede over nine token types incl. the URL and EMAIL expressions -- and the
automaton restatement -- nested pattern references, item order of gatherResultItems -- against output
of the real reference), and the product's compiled lexer tables must report exactly what the oracle
reports.  GPU test: the product end to end."""
import json
import os

import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import rulelang, synth
from tests.l1_table_sim import Tables

HERE = os.path.dirname(os.path.abspath(__file__))


def _case():
    with open(os.path.join(HERE, "golden", "doc_example.json")) as f:
        return json.load(f)


def test_oracle_reproduces_the_published_listings():
    case = _case()
    lx, mt = oracle.L1Lexer(), oracle.L2Matcher()
    prg = rulelang.load(case["program"], lx, mt)
    text = case["text"].encode()
    lex, offs = lx.matchDocs(text, np.array([0, len(text)], np.uint64))
    assert rulelang.format_tokens(prg, text, lex) == case["tokens"]
    res = mt.run(synth.lexems5(lex), offs)
    pn, vn = rulelang.name_tables(prg, mt)
    assert rulelang.format_results(text, res.results, res.items, pn, vn, origin=case["origin"]) == case["results"]


def test_product_tables_report_what_the_oracle_reports():
    case = _case()
    text = case["text"].encode()
    lx, mt = oracle.L1Lexer(), oracle.L2Matcher()
    rulelang.load(case["program"], lx, mt)
    raw, _ = lx.matchDocs(text, [0, len(text)], raw=True)
    plx, pmt = spa.PatternLexerInstance(), spa.PatternMatcherInstance()
    rulelang.load(case["program"], plx, pmt)
    assert Tables(plx.dumpTables()).raw_reports(text) == [(int(r[0]), int(r[1]), int(r[2])) for r in raw]
    # the rule tables of product and oracle are word-for-word the same
    assert list(pmt.dumpTable()) == list(mt.dumpTable())


def test_loader_syntax():
    class Rec:
        def __init__(self):
            self.calls = []

        def __getattr__(self, name):
            return lambda *a: self.calls.append((name,) + a)
    lx, mt = Rec(), Rec()
    prg = rulelang.load('''
        # comment
        WORD ^1 : /\\b\\w+\\b/;
        NUM : @[0-9]+@ | /[0-9]+[.][0-9]+/;
        .Year = sequence_imm( WORD "in", WORD "the", year=NUM | 3 ) ["{year}"];
        Event = within( when=Year, any( WORD "fair", WORD "show" | 1, 1 ) | 10 );
    ''', lx, mt)
    assert [c for c in lx.calls if c[0] == "defineLexem"] == [("defineLexem", 1, "\\b\\w+\\b", 0, 1, "content"),
                                                             ("defineLexem", 2, "[0-9]+", 0, 0, "content"),
                                                             ("defineLexem", 2, "[0-9]+[.][0-9]+", 0, 0, "content")]
    assert [c[1:] for c in lx.calls if c[0] == "defineSymbol"] == [(3, 1, "in"), (4, 1, "the"), (5, 1, "fair"), (6, 1, "show")]
    assert mt.calls == [("pushTerm", 3), ("pushTerm", 4), ("pushTerm", 2), ("attachVariable", "year"), ("pushExpression", "sequence_imm", 3, 3, 0),
                        ("definePattern", "Year", "{year}", False),
                        ("pushPattern", "Year"), ("attachVariable", "when"), ("pushTerm", 5), ("pushTerm", 6), ("pushExpression", "any", 2, 1, 1),
                        ("pushExpression", "within", 2, 10, 0), ("definePattern", "Event", "", True), ("compile",)]
    assert prg.patterns == ["Year", "Event"] and prg.variables == ["year", "when"]
    with pytest.raises(rulelang.RuleLangError):
        rulelang.load("A : /x/ ~2;", Rec(), Rec())
    with pytest.raises(rulelang.RuleLangError):
        rulelang.load("P = sequence( Q, R | 2 );", Rec(), Rec())


@pytest.mark.gpu
def test_product_reproduces_the_published_listings():
    case = _case()
    lx, mt = spa.PatternLexerInstance(), spa.PatternMatcherInstance()
    prg = rulelang.load(case["program"], lx, mt)
    text = case["text"].encode()
    lb = lx.createContext().matchDocs(text, np.array([0, len(text)], np.uint64))
    assert rulelang.format_tokens(prg, text, lb.lexems) == case["tokens"]
    mb = mt.createContext().matchDocs(lb.lexems, lb.doc_offsets)
    assert rulelang.format_results(text, mb.results, mb.items, mt.patternName, mt.variableName, origin=case["origin"]) == case["results"]
