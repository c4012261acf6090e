"""Pins the CPU oracle (oracle/l2_oracle.cpp) against the reference's own known answers."""
import numpy as np

import oracle
from tests import l2_cases


def test_simple_token_pattern_match_golden():
    case = l2_cases.load("simple_token_pattern_match.json")
    m = oracle.L2Matcher()
    l2_cases.build_simple(m, case)
    lex = l2_cases.simple_doc(case)
    res = m.run(lex, [0, len(lex)])
    got = sorted({(int(r[0]), int(r[1])) for r in res.results})
    exp = sorted((m.patternId(r["name"]), p) for r in case["rules"] for p in r["expected_ordpos"])
    assert got == exp
    # 18 programs installed, 26 signals on this document
    assert res.stats[0, 0] == 18 and res.stats[0, 2] == 26


def test_nested_within_sequence_golden():
    case = l2_cases.load("nested_within_sequence.json")
    m = oracle.L2Matcher()
    l2_cases.build_nested(m, case)
    lex = l2_cases.nested_doc(case)
    res = m.run(lex, [0, len(lex)])
    h = m.patternId("outer")
    vid = {m.variableId("b"): "b", m.variableId("c"): "c"}
    assert len(res.results) == len(case["results"])
    for r, e in zip(res.results, case["results"]):
        assert (int(r[0]), int(r[1]), int(r[2]), int(r[4]), int(r[6])) == (h, e["ordpos"], e["ordend"], e["origpos"], e["origend"])
        items = res.items[r[7]:r[7] + r[8]]
        assert [[vid[int(i[0])], int(i[1]), int(i[2])] for i in items] == e["items"]
    assert res.stats[0, 0] == case["programs_installed"]
    assert res.stats[0, 2] == case["signals_fired"]
