"""CPU-side tests of the product's rule compiler (no GPU needed): the flat ProgramTable the
product uploads must equal, word for word, the table the oracle restatement of
src/ruleMatcherAutomaton.cpp:259-586 + src/patternMatcher.cpp:377-584 builds from the same calls."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import capi, synth
from tests import l2_cases


def _both(build):
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    build(m)
    build(o)
    return m.dumpTable(), o.dumpTable()


def test_capi_exports_every_declared_symbol():
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "strus_pattern_amd.h")).read()
    declared = set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", hdr))
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "library does not export " + name
    assert declared <= set(capi.SIGNATURES), sorted(declared - set(capi.SIGNATURES))


def test_golden_tables():
    case = l2_cases.load("simple_token_pattern_match.json")
    a, b = _both(lambda m: l2_cases.build_simple(m, case))
    assert np.array_equal(a, b)
    case = l2_cases.load("nested_within_sequence.json")
    a, b = _both(lambda m: l2_cases.build_nested(m, case))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("nrules,nfeat,seed,op", [
    (10000, 10000, 7, None), (3000, 300, 8, "sequence"), (2000, 50, 9, None), (500, 20, 10, "within_struct"),
    (1000, 100, 11, "any"), (1000, 100, 12, "sequence_struct"),
])
def test_random_rule_tables(nrules, nfeat, seed, op):
    rules = synth.random_rules(nrules, nfeat, seed, op)
    a, b = _both(lambda m: synth.apply_rules(m, rules))
    assert np.array_equal(a, b)


def test_optimizer_options_and_frequencies():
    rules = synth.random_rules(3000, 200, 13)

    def build(m):
        m.defineOption("stopwordOccurrenceFactor", 0.002)
        m.defineOption("weightFactor", 2.0)
        m.defineOption("maxRange", 8)
        for t in range(1, 50):
            m.defineTermFrequency(t, 1.0 + (t % 7))
        synth.apply_rules(m, rules)

    a, b = _both(build)
    assert np.array_equal(a, b)
    assert a[2] > 0  # some stop words


def test_error_behaviour_matches_reference():
    m = spa.PatternMatcherInstance()
    with pytest.raises(spa.PatternError):
        m.attachVariable("x")            # no node on the stack (patternMatcher.cpp:527-530)
    with pytest.raises(spa.PatternError):
        m.pushExpression("sequence", 2, 1, 0)   # more arguments than nodes (:388-391)
    m.pushTerm(1)
    m.attachVariable("a")
    with pytest.raises(spa.PatternError):
        m.attachVariable("b")            # more than one variable (:532-535)
    with pytest.raises(spa.PatternError):
        m.defineOption("nonsense", 1.0)  # unknown option (:638-641)
    for _ in range(33):
        m.pushTerm(2)
    with pytest.raises(spa.PatternError):
        m.pushExpression("within", 33, 5, 0)    # within arity > 32 (:418-421)
    with pytest.raises(spa.PatternError):
        m.defineTermFrequency(1, 0.0)    # df must be positive (ruleMatcherAutomaton.cpp:261-264)


def test_context_requires_gpu_or_fails_loudly():
    """No CPU fallback: without a usable HIP device createContext must raise."""
    if spa.device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    m = spa.PatternMatcherInstance()
    m.pushTerm(1)
    m.definePattern("p")
    with pytest.raises(spa.PatternError):
        m.createContext()


def test_fast_tier_eligibility():
    """which compiled rule sets run on the LDS-resident kernel (struspattern_amd/csrc/l2_fast_tables.cpp)"""
    import struspattern_amd as spa
    from struspattern_amd import synth

    def inst(build):
        m = spa.PatternMatcherInstance()
        build(m)
        return m.fastTier()

    assert inst(lambda m: synth.apply_rules(m, synth.random_rules(3000, 500, 7)))[0]
    assert inst(lambda m: synth.apply_rules(m, synth.random_rules(300, 50, 8), compile=False))[0]

    def nested(m):
        m.pushTerm(1); m.pushTerm(2); m.pushExpression("sequence", 2, 3, 0)
        m.pushTerm(3); m.pushExpression("within", 2, 5, 0); m.definePattern("b", "", True); m.compile()
    ok, why = inst(nested)
    assert not ok and "nested" in why

    def conj(m):
        m.pushTerm(1); m.pushTerm(2); m.pushExpression("and", 2, 3, 0); m.definePattern("a", "", True)
    assert not inst(conj)[0]

    def far(m):
        m.pushTerm(1); m.pushTerm(2); m.pushExpression("sequence", 2, 64, 0); m.definePattern("a", "", True)
    assert not inst(far)[0]

    def wide(m):
        for t in (1, 2, 3, 4):
            m.pushTerm(t)
        m.pushExpression("sequence", 4, 9, 0); m.definePattern("a", "", True)
    assert not inst(wide)[0]

    def near(m):
        m.pushTerm(1); m.pushTerm(2); m.pushTerm(3); m.pushExpression("within", 3, 63, 0); m.definePattern("a", "", True)
    assert inst(near)[0]
