"""Shared L2 test inputs: the reference's known-answer cases restated as data + builders that
drive ANY object with the PatternMatcherInstance method names (oracle.L2Matcher or the product's
struspattern_amd.PatternMatcherInstance)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def simple_doc(case):
    """testSimpleTokenPatternMatch.cpp:79-98 -> (n,5) lexems [id, ordpos, origseg, origpos, origsize]."""
    items = []
    for ii in range(case["doc_size"]):
        items.append((ii + 1, ii + 1))
        if (ii + 1) % 10 == 0:
            items.append((case["delim"], ii + 2))
    ii = case["doc_size"]
    items.append((1, ii + 1))
    items.append((1, ii + 2))
    lex = np.zeros((len(items), 5), np.uint32)
    for idx, (tid, pos) in enumerate(items):
        lex[idx] = (tid, pos, 0, idx, 1)
    return lex


def build_simple(m, case):
    """createPattern/createPatterns of the same test (:121-157); compile() at :244."""
    for r in case["rules"]:
        for termid, var in r["terms"]:
            m.pushTerm(termid)
            m.attachVariable(var)
        m.pushExpression(r["op"], len(r["terms"]), r["range"], r["cardinality"])
        m.definePattern(r["name"], "", not r["name"].startswith("_"))
    m.compile()


def build_nested(m, case):
    t = case["terms"]
    m.pushTerm(t["A"])
    m.pushTerm(t["B"])
    m.attachVariable("b")
    m.pushExpression("sequence", 2, 2, 0)
    m.pushTerm(t["C"])
    m.attachVariable("c")
    m.pushExpression("within", 2, 6, 0)
    m.definePattern("outer", "", True)
    m.compile()


def nested_doc(case):
    t = case["terms"]
    lex = np.zeros((len(case["tokens"]), 5), np.uint32)
    for i, tok in enumerate(case["tokens"]):
        lex[i] = (t[tok], i + 1, 0, i, 1)
    return lex
