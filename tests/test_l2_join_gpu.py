"""The opt-in join prototype of the rule automaton (csrc/l2_join.h, SPA_L2_JOIN=1): two-term rule sets (sequence, within,
sequence_struct, within_struct, any: the operators of the reference's randomTokenPatternMatch) evaluated without
materialised rule instances.  It reproduces the reference's result SETS per document, not the order
of the results inside a document and not the statistics -- so this test compares sorted result tuples with the
oracle (unoptimized automaton), and checks that the default engine is untouched by the switch."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth

pytestmark = pytest.mark.gpu


def _docs(rng, ndocs, n, nfeat, shared_positions):
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    for d in range(ndocs):
        ids = rng.integers(1, nfeat + 1, size=n)
        ids[rng.random(n) < 0.06] = synth.DELIM
        pos = np.arange(1, n + 1)
        if shared_positions:
            pos = np.cumsum(rng.random(n) < 0.7) + 1          # several lexems on one position
        lex[d * n:(d + 1) * n, 0] = ids
        lex[d * n:(d + 1) * n, 1] = pos
        lex[d * n:(d + 1) * n, 2] = np.arange(n) * 3
        lex[d * n:(d + 1) * n, 3] = 2
    return lex, offs


def _sorted_results(batch, d):
    """the results of document d with their captured items (variable, positions; in the result's list order), sorted"""
    out = []
    for r in batch.results[batch.doc_offsets[d]:batch.doc_offsets[d + 1]].tolist():
        it = batch.items[r[7]:r[7] + r[8]]
        out.append(tuple(r[:7]) + (r[8],) + tuple(it.reshape(-1).tolist()))
    return sorted(out)


@pytest.mark.parametrize("op", ["sequence", "within", "sequence_struct", "within_struct", "any", None])
@pytest.mark.parametrize("seed,nrules,nfeat,n,shared", [(1, 50, 8, 300, False), (2, 400, 30, 500, True), (3, 3000, 200, 1000, True), (4, 20, 3, 200, True)])
def test_join_prototype_result_sets(monkeypatch, seed, nrules, nfeat, n, shared, op):
    rng = np.random.default_rng(100 + seed)
    rules = synth.random_rules(nrules, nfeat, seed, op=op)
    monkeypatch.setenv("SPA_L2_JOIN", "1")
    m = spa.PatternMatcherInstance()
    synth.apply_rules(m, rules, compile=False)        # (not optimized: no alternative keys)
    ctx = m.createContext()
    assert ctx.kernelKind() == 2
    monkeypatch.delenv("SPA_L2_JOIN")
    o = oracle.L2Matcher()
    synth.apply_rules(o, rules, compile=False)
    ndocs = 24
    lex, offs = _docs(rng, ndocs, n, nfeat, shared)
    got = ctx.matchDocs(lex, offs)
    ref = o.run(synth.lexems5(lex), offs)
    assert np.array_equal(got.doc_offsets, ref.doc_offsets)
    total = 0
    for d in range(ndocs):
        a = _sorted_results(got, d)
        b = _sorted_results(ref, d)
        assert a == b, (d, len(a), len(b), [x for x in a if x not in b][:2], [x for x in b if x not in a][:2])
        total += len(a)
    assert total > 100
    assert len(got.items) == len(ref.items) > 0
    # the same instance without the switch runs the exact engine
    assert m.createContext().kernelKind() == 1


def test_join_prototype_leaves_other_rule_sets_to_the_exact_engine(monkeypatch):
    monkeypatch.setenv("SPA_L2_JOIN", "1")
    m = spa.PatternMatcherInstance()
    m.pushTerm(1); m.pushTerm(2); m.pushTerm(3)
    m.pushExpression("sequence", 3, 10, 0)                       # three terms
    m.definePattern("x", "", True)
    assert m.createContext().kernelKind() != 2


def test_join_prototype_on_the_headline_rule_set(monkeypatch):
    """The 10k + 10k pipeline workload of bench.py: the join prototype's result sets against the exact engine's (same rule
    set, not optimized -- the optimizer's alternative keys change ~0.02 % of the results of the exact engine itself)."""
    vocab = synth.vocabulary(30000, 1)
    pats, rules = synth.pipeline_workload(10000, 10000, vocab, seed=4)
    text, offs = synth.text_documents(48, 16384, vocab, seed=1000, utf8=True)
    lxi = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lxi, pats)
    lex = lxi.createContext().matchDocs(text, offs)
    exact = spa.PatternMatcherInstance()
    synth.apply_rules(exact, rules, compile=False)
    monkeypatch.setenv("SPA_L2_JOIN", "1")
    ctx = exact.createContext()
    assert ctx.kernelKind() == 2
    monkeypatch.delenv("SPA_L2_JOIN")
    ectx = exact.createContext()
    assert ectx.kernelKind() == 1
    got = ctx.matchDocs(lex.lexems, lex.doc_offsets)
    ref = ectx.matchDocs(lex.lexems, lex.doc_offsets)
    assert np.array_equal(got.doc_offsets, ref.doc_offsets) and len(ref.results) > 100000
    assert len(got.items) == len(ref.items) > 100000
    for d in range(len(offs) - 1):
        assert _sorted_results(got, d) == _sorted_results(ref, d), d
