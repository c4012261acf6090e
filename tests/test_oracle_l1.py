"""Pins the level-1 CPU oracle: (1) the reference's own 36-lexem known answer (the only vector
that ties it to real Hyperscan), (2) a differential check of the regex semantics against Python's
`re`, (3) unit checks of the handler restatement."""
import random
import re

import numpy as np
import pytest

import oracle
from tests import l1_cases


def test_char_regex_match_golden_case1():
    case = l1_cases.load_char_regex_cases()[0]
    lx = oracle.L1Lexer()
    l1_cases.build_case(lx, case)
    got = lx.match(case["src"].encode()).tolist()
    assert got == case["result"]


@pytest.mark.parametrize("index", [1, 2])
def test_char_regex_match_golden_edit_distance_cases(index):
    """testCharRegexMatch.cpp:161-196: `abc ~1` on ASCII and `a\u00f6\u00fc ~1` on UTF-8 text, 5 lexems each -- the two
    vectors that pin the restatement of the approximate route (oracle/l1_oracle.cpp)."""
    case = l1_cases.load_char_regex_cases()[index]
    lx = oracle.L1Lexer()
    l1_cases.build_case(lx, case)
    got = lx.match(case["src"].encode()).tolist()
    assert got == case["result"]


def test_edit_distance_outside_the_restated_part_is_rejected_not_silently_ignored():
    for expr in ("a[bc]d ~1", "\\bword\\b ~1", "ab ~2"):
        lx = oracle.L1Lexer()
        lx.defineLexem(1, expr, 0, 1, "content")
        with pytest.raises(oracle.OracleError):
            lx.compile()


@pytest.mark.parametrize("seed", range(6))
def test_report_semantics_against_python_re(seed):
    """every end offset once, leftmost start, no empty matches."""
    rng = random.Random(seed)
    for _ in range(60):
        pat = l1_cases.random_regex(rng)
        try:
            re.compile(pat)
        except re.error:
            continue
        text = l1_cases.random_text(rng, rng.randint(0, 24))
        lx = oracle.L1Lexer()
        lx.defineOption("DOTALL")
        lx.defineLexem(1, pat, 0, 1, "content")
        try:
            lx.compile()
        except oracle.OracleError as e:         # (an expression that matches the empty buffer needs ALLOWEMPTY, as with Hyperscan)
            assert "matches empty buffer" in str(e), str(e)
            # (Python's \\B never matches in an empty string, PCRE's does: no cross-check for such expressions)
            assert "\\B" in pat or re.compile(pat, re.DOTALL | re.ASCII).fullmatch("") is not None, pat
            continue
        assert "\\B" in pat or re.compile(pat, re.DOTALL | re.ASCII).fullmatch("") is None, pat
        raw, _ = lx.matchDocs(text.encode(), [0, len(text)], raw=True)
        got = [(int(r[1]), int(r[2])) for r in raw]
        exp = l1_cases.py_leftmost_reports(pat, text, re.DOTALL | re.ASCII)
        assert got == exp, (pat, text)


def test_options_caseless_multiline_dot():
    def reports(pat, text, *opts):
        lx = oracle.L1Lexer()
        for o in opts:
            lx.defineOption(o)
        lx.defineLexem(1, pat, 0, 1, "content")
        lx.compile()
        raw, _ = lx.matchDocs(text.encode(), [0, len(text.encode())], raw=True)
        return [(int(r[1]), int(r[2])) for r in raw]

    assert reports("abc", "xABCx", "CASELESS") == [(1, 4)]
    assert reports("abc", "xABCx") == []
    assert reports("a.c", "a\nc") == []
    assert reports("a.c", "a\nc", "DOTALL") == [(0, 3)]
    assert reports("^b", "a\nb", "MULTILINE") == [(2, 3)]
    assert reports("^b", "a\nb") == []
    assert reports("a$", "a\nb", "MULTILINE") == [(0, 1)]
    # UTF-8: '.' and negated classes consume whole code points
    assert reports("a.c", "aöc") == [(0, 4)]
    assert reports("[^x]+", "ö") == [(0, 2)]
    assert reports("ö+", "aöö") == [(1, 3), (1, 5)]
    assert reports("[ä-ü]", "ö") == [(0, 2)]


def test_supersede_levels_and_ordpos():
    """handler + ordinal positions (patternLexer.cpp:717-826, :893-945) on a hand-made case:
    a higher-level lexem covering lower-level ones removes them; a covered lower-level one is ignored."""
    lx = oracle.L1Lexer()
    lx.defineLexem(1, "\\b\\w+\\b", 0, 1, "content")          # WORD ^1
    lx.defineLexem(2, "[.]", 0, 2, "content")                 # SENT ^2
    lx.defineLexem(3, "\\b[a-z]+[.][a-z]+\\b", 0, 3, "content")  # DOTTED ^3 covers WORD . WORD
    lx.compile()
    got = lx.match(b"go to example.com now.").tolist()
    assert got == [[1, 1, 0, 2], [1, 2, 3, 2], [3, 3, 6, 11], [1, 4, 18, 3], [2, 5, 21, 1]]


def test_symbols_emit_a_twin_lexem():
    lx = oracle.L1Lexer()
    lx.defineLexem(1, "[a-z]+\\b", 0, 1, "content")
    lx.defineSymbol(7, 1, "cat")
    lx.compile()
    assert lx.getSymbol(1, "cat") == 7 and lx.getSymbol(1, "dog") == 0
    got = lx.match(b"a cat sat").tolist()
    assert got == [[1, 1, 0, 1], [1, 2, 2, 3], [7, 2, 2, 3], [1, 3, 6, 3]]
