"""Result format strings (patternMatcher.cpp:172-181, :253-262, :561-566).

What the engine contributes -- which format handle a result / item carries and which records are the
arguments of that format -- is compared with the oracle bit for bit.  The formatter itself is part of
strusAnalyzer (not in the reference repository): struspattern_amd/resultformat.py implements the
documented behaviour and is pinned only by the one worked example of the reference's web page
(doc/webpage/introduction_struspattern.htm:143-161: "3'645 Eur" -> "EUR 3'645")."""
import random

import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import resultformat, rulelang, synth

# the program of introduction_struspattern.htm:146-156 (+ the private-pattern example of :138-141); the
# position ranges are added (the page's loader, part of strusAnalyzer, is not in the reference repository)
MONEY = r'''
    WORD ^1         : /\b\w+\b/;
    NUMBER ^2       : /\b[0-9]{4}\b/;
    AMOUNT^5        : /\b[0-9]{1,3}'[0-9]{3,3}'[0-9]{3,3}\b/;
    AMOUNT^4        : /\b[0-9]{1,3}'[0-9]{3,3}\b/;
    AMOUNT^3        : /\b[0-9]{1,3}\b/;
    CURRENCY_CHF ^3 : /\b[Ss]{0,1}[Ff][Rr][.]{0,1}\b/;
    CURRENCY_EUR ^3 : /\b[Ee][Uu][Rr][.]{0,1}\b/;
    Currency        = CURRENCY_CHF ["CHF"];
    Currency        = CURRENCY_EUR ["EUR"];
    MoneyAmount     = sequence_imm( value=AMOUNT, currency=Currency | 2) ["{currency} {value}"];
    .Year           = sequence_imm( WORD "in", WORD "the", WORD "year", year=NUMBER | 4) ["{year}"];
    Event           = sequence_imm( when=Year, WORD "something", WORD "happened" | 8) ["{when}"];
    Plain           = sequence( when=Year, what=WORD | 8 );
'''
TEXT = b"He paid 3'645 Eur and 12 sFr. in the year 1984 something happened again"


def _run_oracle(program, text):
    lx, mt = oracle.L1Lexer(), oracle.L2Matcher()
    prg = rulelang.load(program, lx, mt)
    lex, offs = lx.matchDocs(text, np.array([0, len(text)], np.uint64))
    res = mt.run(synth.lexems5(lex), offs)
    return prg, mt, res


def _listing(fm, res, text):
    out = []
    for r in fm.results(res.results, res.items, res.result_format, res.item_format, text):
        out.append((r.name, r.value, r.text(text), [(i.name, i.value, i.text(text)) for i in r.items]))
    return out


def test_format_string_parser():
    assert resultformat.parse("CHF") == ["CHF"]
    assert resultformat.parse("{currency} {value}") == [("currency", " "), " ", ("value", " ")]
    assert resultformat.parse("a{x|, }b\\{c") == ["a", ("x", ", "), "b{c"]
    with pytest.raises(resultformat.FormatError):
        resultformat.parse("{open")
    with pytest.raises(resultformat.FormatError):
        resultformat.parse("{}")


def test_documented_example_on_the_oracle():
    prg, mt, res = _run_oracle(MONEY, TEXT)
    assert prg.formats == ["CHF", "EUR", "{currency} {value}", "{year}", "{when}"]
    got = _listing(rulelang.formatter(prg, mt), res, TEXT)
    # the web page's example: "3'645 Eur" generates the value "EUR 3'645"
    assert ("MoneyAmount", "EUR 3'645", "3'645 Eur", []) in got
    assert ("MoneyAmount", "CHF 12", "12 sFr", []) in got
    assert ("Currency", "EUR", "Eur", []) in got and ("Currency", "CHF", "sFr", []) in got
    # a format string on a sub-pattern is the value of the variable bound to it; its own variables are consumed
    assert ("Event", "1984", "in the year 1984 something happened", []) in got
    # without a format string the result keeps its items (latest first); `year`, the argument of Year's format, is not among them
    assert ("Plain", None, "in the year 1984 something", [("what", None, "something"), ("when", "1984", "in the year 1984")]) in got
    assert not [g for g in got if g[0] == "Year"]


def _random_program(rng, nterms):
    """token rules with private and public sub-patterns, some with format strings, referenced with variables"""
    calls, names = [], []
    for k in range(14):
        name = "P%d" % k

        def operand(depth=0):
            if names and rng.random() < (0.45 if depth == 0 else 0.25):
                calls.append(("pushPattern", rng.choice(names)))
            elif depth < 2 and rng.random() < 0.2:
                argc = rng.randint(2, 3)
                for _ in range(argc):
                    operand(depth + 1)
                calls.append(("pushExpression", rng.choice(["sequence", "within", "any"]), argc, rng.randint(3, 8), 0))
            else:
                calls.append(("pushTerm", rng.randint(1, nterms)))
            if rng.random() < 0.7:
                calls.append(("attachVariable", "v%d" % rng.randint(0, 3)))
        argc = rng.randint(1, 3)
        for _ in range(argc):
            operand()
        calls.append(("pushExpression", rng.choice(["sequence", "sequence_imm", "within", "any", "sequence_struct"]) if argc > 1 else "any", argc, rng.randint(2, 10), 0))
        fmt = rng.choice(["", "", "<{v0}>", "{v1|,}+{v2}", "const", "{v0}{v3}"])
        calls.append(("definePattern", name, fmt, rng.random() < 0.7))
        names.append(name)
    return calls


def _apply(m, calls):
    for c in calls:
        if c[0] == "pushExpression" and c[1] == "sequence_struct":
            m.pushTerm(1)   # the delimiter operand comes first
            getattr(m, c[0])(c[1], c[2] + 1, c[3], c[4])
            continue
        getattr(m, c[0])(*c[1:])
    m.compile()


@pytest.mark.gpu
def test_documented_example_on_the_gpu():
    prg, omt, ref = _run_oracle(MONEY, TEXT)
    lx, mt = spa.PatternLexerInstance(), spa.PatternMatcherInstance()
    prg2 = rulelang.load(MONEY, lx, mt)
    assert [mt.formatString(h + 1) for h in range(mt.formatCount())] == prg.formats == prg2.formats
    lb = lx.createContext().matchDocs(TEXT, np.array([0, len(TEXT)], np.uint64))
    mb = mt.createContext().matchDocs(lb.lexems, lb.doc_offsets)
    assert np.array_equal(mb.results, ref.results) and np.array_equal(mb.items, ref.items)
    assert np.array_equal(mb.result_format, ref.result_format) and np.array_equal(mb.item_format, ref.item_format)
    assert _listing(rulelang.formatter(prg2, mt), mb, TEXT) == _listing(rulelang.formatter(prg, omt), ref, TEXT)
    # single-document interface
    ctx = mt.createContext()
    for lid, ordpos, origpos, origsize in lb.lexems.tolist():
        ctx.putInput(lid, ordpos, origpos, origsize)
    r, it = ctx.fetchResults()
    rf, itf = ctx.fetchFormats(len(r), len(it))
    assert np.array_equal(r, ref.results) and np.array_equal(it, ref.items)
    assert np.array_equal(rf, ref.result_format) and np.array_equal(itf, ref.item_format)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_programs_with_format_strings(seed):
    rng = random.Random(9100 + seed)
    nterms = 6
    calls = _random_program(rng, nterms)
    mt, omt = spa.PatternMatcherInstance(), oracle.L2Matcher()
    _apply(mt, calls)
    _apply(omt, calls)
    assert list(mt.dumpTable()) == list(omt.dumpTable())
    lex, offs = synth.random_documents(40, 120, nterms, seed=50 + seed)
    gpu = mt.createContext().matchDocs(lex, offs)
    ref = omt.run(synth.lexems5(lex), offs, nthreads=4)
    assert len(ref.results) > 50 and ref.item_format is not None and (int(ref.item_format[:, 0].max()) > 0 or int(ref.result_format.max()) > 0)
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results, ref.results)
    assert np.array_equal(gpu.items, ref.items)
    assert np.array_equal(gpu.result_format, ref.result_format)
    assert np.array_equal(gpu.item_format, ref.item_format)
    assert np.array_equal(gpu.stats, ref.stats)


@pytest.mark.gpu
def test_matcher_without_format_strings_reports_none():
    mt = spa.PatternMatcherInstance()
    mt.pushTerm(1); mt.attachVariable("a"); mt.pushTerm(2); mt.pushExpression("sequence", 2, 3, 0); mt.definePattern("p", "", True)
    mt.compile()
    lex, offs = synth.random_documents(4, 50, 3, seed=1)
    b = mt.createContext().matchDocs(lex, offs)
    assert b.result_format is None and b.item_format is None and len(b.results) > 0
