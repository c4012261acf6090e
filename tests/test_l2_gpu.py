"""GPU parity tests of the level-2 rule automaton: HIP kernel (through the C-ABI) vs the CPU oracle
on the same inputs.  Bit-exact: per document the result 7-tuples in firing order, the captured
items in list order, and the automaton statistics."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth
from tests import l2_cases

pytestmark = pytest.mark.gpu


def _compare(gpu, ref, ndocs, with_items=True):
    assert np.array_equal(gpu.status, np.zeros(ndocs, np.int32))
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results[:, :7], ref.results[:, :7])
    assert np.array_equal(gpu.stats, ref.stats)
    if with_items:
        assert np.array_equal(gpu.results[:, 8], ref.results[:, 8])
        assert np.array_equal(gpu.items, ref.items)


def _run_both(build, lex4, offs, origseg=None):
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    build(m)
    build(o)
    ctx = m.createContext()
    gpu = ctx.matchDocs(lex4, offs, origseg)
    l5 = synth.lexems5(lex4)
    if origseg is not None:
        l5[:, 2] = origseg
    ref = o.run(l5, offs)
    return gpu, ref, m, o


def test_simple_token_pattern_match_golden():
    case = l2_cases.load("simple_token_pattern_match.json")
    lex5 = l2_cases.simple_doc(case)
    lex4 = lex5[:, [0, 1, 3, 4]]
    gpu, ref, m, o = _run_both(lambda x: l2_cases.build_simple(x, case), lex4, [0, len(lex4)])
    got = sorted({(m.patternName(int(r[0])), int(r[1])) for r in gpu.results})
    exp = sorted((r["name"], p) for r in case["rules"] for p in r["expected_ordpos"])
    assert got == exp
    _compare(gpu, ref, 1)


def test_nested_within_sequence_golden():
    case = l2_cases.load("nested_within_sequence.json")
    lex5 = l2_cases.nested_doc(case)
    lex4 = lex5[:, [0, 1, 3, 4]]
    gpu, ref, m, o = _run_both(lambda x: l2_cases.build_nested(x, case), lex4, [0, len(lex4)])
    assert len(gpu.results) == len(case["results"])
    for r, e in zip(gpu.results, case["results"]):
        assert (int(r[1]), int(r[2]), int(r[4]), int(r[6])) == (e["ordpos"], e["ordend"], e["origpos"], e["origend"])
        items = gpu.items[r[7]:r[7] + r[8]]
        assert [[m.variableName(int(i[0])), int(i[1]), int(i[2])] for i in items] == e["items"]
    assert gpu.stats[0, 0] == case["programs_installed"] and gpu.stats[0, 2] == case["signals_fired"]
    _compare(gpu, ref, 1)


@pytest.mark.parametrize("nrules,nfeat,ndocs,docsize,seed,op,optimize", [
    (2000, 200, 300, 300, 21, None, True),
    (2000, 200, 300, 300, 22, None, False),
    (5000, 1000, 200, 1000, 23, "sequence", True),
    (1000, 50, 100, 500, 24, "within", True),
    (1000, 50, 100, 500, 25, "sequence_struct", True),
    (1000, 50, 100, 500, 26, "within_struct", True),
    (500, 30, 100, 300, 27, "any", True),
])
def test_random_token_pattern_match(nrules, nfeat, ndocs, docsize, seed, op, optimize):
    rules = synth.random_rules(nrules, nfeat, seed, op)
    lex, offs = synth.random_documents(ndocs, docsize, nfeat, seed + 1000)
    gpu, ref, m, o = _run_both(lambda x: synth.apply_rules(x, rules, compile=optimize), lex, offs)
    assert len(ref.results) > 0
    _compare(gpu, ref, ndocs)


def test_edge_cases_empty_and_ragged_documents():
    rules = synth.random_rules(300, 20, 31)
    lex, offs = synth.random_documents(50, 40, 20, 32)
    # make ragged: empty documents at the start, in the middle and at the end, and a one-lexem document
    offs2 = np.concatenate([[0, 0], offs[:20], [offs[19]], offs[20:22], [offs[21] + 1], offs[22:], [offs[-1]]]).astype(np.uint64)
    offs2 = np.maximum.accumulate(offs2)
    gpu, ref, m, o = _run_both(lambda x: synth.apply_rules(x, rules), lex, offs2)
    _compare(gpu, ref, len(offs2) - 1)


def test_origseg_and_position_gaps():
    """ordinal positions with jumps > 64 exercise the far-expiry heap (ruleMatcherAutomaton.cpp:1093-1129)."""
    rng = np.random.default_rng(41)
    rules = []
    for ni in range(400):
        rg = int(rng.integers(1, 200))
        rules.append(("r%d" % ni, ["sequence", "within", "sequence_struct", "within_struct"][ni % 4], rg,
                      [int(rng.integers(1, 12)), int(rng.integers(1, 12))]))
    ndocs, n = 60, 400
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    seg = np.zeros(ndocs * n, np.uint32)
    for d in range(ndocs):
        steps = rng.choice([0, 1, 1, 1, 2, 5, 63, 64, 65, 130, 300], size=n)
        steps[0] = 1
        pos = np.cumsum(steps)
        ids = rng.integers(1, 12, size=n)
        ids[rng.random(n) < 0.08] = synth.DELIM
        lex[d * n:(d + 1) * n, 0] = ids
        lex[d * n:(d + 1) * n, 1] = pos
        lex[d * n:(d + 1) * n, 2] = np.arange(n) * 3
        lex[d * n:(d + 1) * n, 3] = 2
        seg[d * n:(d + 1) * n] = np.arange(n) // 100
    gpu, ref, m, o = _run_both(lambda x: synth.apply_rules(x, rules), lex, offs, seg)
    assert len(ref.results) > 0
    _compare(gpu, ref, ndocs)


def _apply_wide_rules(m, rules, nested):
    """rules with many terms: more than 4 installed triggers per rule instance (continuation blocks of
    the kernel's rule records) and more than 3 trigger templates (sequential install path)."""
    for name, op, rg, card, params in rules:
        n = len(params)
        if op in ("sequence_struct", "within_struct"):
            m.pushTerm(synth.DELIM)
            n += 1
        for pi, t in enumerate(params):
            m.pushTerm(t)
            if pi % 2 == 0:
                m.attachVariable("V%d" % pi)
        m.pushExpression(op, n, rg, card)
        m.definePattern(name, "", True)
    for name, inner, t in nested:
        # a pattern reference inside a wide expression: follow events + joined item lists
        m.pushPattern(inner)
        m.attachVariable("sub")
        m.pushTerm(t)
        m.pushExpression("sequence", 2, 12, 0)
        m.definePattern(name, "", True)
    m.compile()


@pytest.mark.parametrize("seed", [51, 52])
def test_rules_with_many_triggers(seed):
    rng = np.random.default_rng(seed)
    nfeat = 9
    ops = ["sequence", "within", "any", "sequence_struct", "within_struct", "and"]
    rules = []
    for ni in range(160):
        op = ops[ni % len(ops)]
        nterms = int(rng.integers(5, 10))
        params = [int(x) for x in rng.integers(1, nfeat + 1, size=nterms)]
        rg = int(rng.integers(6, 40))
        card = 0
        if op in ("any", "and") and rng.random() < 0.5:
            card = int(rng.integers(2, 4))
        rules.append(("w%s_%d" % (op, ni), op, rg, card, params))
    nested = [("n_%d" % i, rules[i][0], int(rng.integers(1, nfeat + 1))) for i in range(0, 40, 3)]
    ndocs, n = 24, 300
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    for d in range(ndocs):
        ids = rng.integers(1, nfeat + 1, size=n)
        ids[rng.random(n) < 0.04] = synth.DELIM
        lex[d * n:(d + 1) * n, 0] = ids
        lex[d * n:(d + 1) * n, 1] = np.cumsum(rng.choice([1, 1, 1, 2], size=n))
        lex[d * n:(d + 1) * n, 2] = np.arange(n) * 4
        lex[d * n:(d + 1) * n, 3] = 3
    gpu, ref, m, o = _run_both(lambda x: _apply_wide_rules(x, rules, nested), lex, offs)
    assert len(ref.results) > 100
    _compare(gpu, ref, ndocs)


def _random_tree(rng, nfeat, depth, maxdepth):
    """(range, push function) of a random expression tree (BASELINE.json configs[3]: nested
    within/sequence expressions; generator after testRandomExpressionTreeMatch.cpp:239-338 with the
    depth limit lifted: range = argc + random + sum of the children's ranges)."""
    if depth >= maxdepth or (depth > 0 and rng.random() < 0.35):
        t = int(rng.integers(1, nfeat + 1))
        var = "t%d" % depth if rng.random() < 0.3 else None

        def push(m, t=t, var=var):
            m.pushTerm(t)
            if var:
                m.attachVariable(var)
        return 1, push
    op = ["sequence", "within", "sequence_struct", "within_struct", "any", "sequence_imm", "and"][int(rng.integers(0, 7))]
    argc = int(rng.integers(2, 4))
    children = [_random_tree(rng, nfeat, depth + 1, maxdepth) for _ in range(argc)]
    rg = argc + int(rng.integers(0, 4)) + sum(c[0] for c in children)
    card = int(rng.integers(1, argc + 1)) if op in ("any", "and") and rng.random() < 0.4 else 0
    var = "e%d" % depth if depth > 0 and rng.random() < 0.3 else None

    def push(m):
        n = argc
        if op in ("sequence_struct", "within_struct"):
            m.pushTerm(synth.DELIM)
            n += 1
        for _, c in children:
            c(m)
        m.pushExpression(op, n, rg, card)
        if var:
            m.attachVariable(var)
    return rg, push


@pytest.mark.parametrize("seed,maxdepth,ntrees,n,nfeat,ndocs", [
    (61, 3, 60, 250, 6, 16), (62, 5, 60, 250, 6, 16), (63, 8, 16, 120, 6, 16),
    (64, 8, 1000, 200, 60, 32),         # BASELINE.json configs[3] shape: depth-8 trees, a thousand of them over a mid-size alphabet
    (66, 8, 10000, 300, 2000, 16)])     # ... and at its tree count (10 000 trees)
def test_random_expression_trees(seed, maxdepth, ntrees, n, nfeat, ndocs):
    rng = np.random.default_rng(seed)
    trees = [_random_tree(rng, nfeat, 0, maxdepth)[1] for _ in range(ntrees)]

    def build(m):
        for i, push in enumerate(trees):
            push(m)
            m.definePattern("tree_%d" % i, "", True)
        m.compile()
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    for d in range(ndocs):
        ids = rng.integers(1, nfeat + 1, size=n)
        ids[rng.random(n) < 0.05] = synth.DELIM
        lex[d * n:(d + 1) * n, 0] = ids
        lex[d * n:(d + 1) * n, 1] = np.arange(1, n + 1)
        lex[d * n:(d + 1) * n, 2] = np.arange(n) * 2
        lex[d * n:(d + 1) * n, 3] = 1
    gpu, ref, m, o = _run_both(build, lex, offs)
    assert len(ref.results) > 50
    _compare(gpu, ref, ndocs)


def test_not_ascending_positions_is_an_error():
    m = spa.PatternMatcherInstance()
    m.pushTerm(1)
    m.definePattern("p")
    ctx = m.createContext()
    lex = np.array([[1, 5, 0, 1], [1, 4, 1, 1]], np.uint32)
    b = ctx.matchDocs(lex, [0, 2], check=False)
    assert b.status[0] == 1
    with pytest.raises(spa.PatternError):
        ctx.matchDocs(lex, [0, 2])


def test_failure_beside_a_document_that_is_run_again(monkeypatch):
    """One document with descending positions and one whose working set exceeds the (small) arena in the same batch: the
    second is run again with a larger arena and succeeds, the first stays failed -- the batch must still report a failure
    (the partial rerun counts failures among the documents it ran again only)."""
    monkeypatch.setenv("SPA_L2_FAST", "0")              # the general kernel: its arena is what overflows
    rules = synth.random_rules(400, 30, 93)
    lex, offs = synth.random_documents(6, 300, 30, 94)
    lex = lex.copy()
    a, b = int(offs[2]), int(offs[3])
    lex[a + 10, 1], lex[a + 11, 1] = lex[a + 11, 1] + 5, lex[a + 10, 1]        # document 2: positions not ascending
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    synth.apply_rules(m, rules)
    synth.apply_rules(o, rules)
    ctx = m.createContext()
    ctx.setArena(max_rules=32, max_triggers=32, bucket_capacity=8, max_items=32, max_follow=8)
    got = ctx.matchDocs(lex, offs, check=False)
    assert got.status[2] == 1 and all(got.status[d] == 0 for d in (0, 1, 3, 4, 5))
    assert ctx.batchCounters()["failed_docs"] == 1
    with pytest.raises(spa.PatternError):
        ctx.matchDocs(lex, offs)
    good = [d for d in range(6) if d != 2]
    sel = np.concatenate([lex[int(offs[d]):int(offs[d + 1])] for d in good])
    soffs = np.cumsum([0] + [int(offs[d + 1] - offs[d]) for d in good]).astype(np.uint64)
    ref = o.run(synth.lexems5(sel), soffs)
    for k, d in enumerate(good):
        r = got.results[int(got.doc_offsets[d]):int(got.doc_offsets[d + 1]), :7]
        e = ref.results[int(ref.doc_offsets[k]):int(ref.doc_offsets[k + 1]), :7]
        assert np.array_equal(r, e), "document %d" % d


def test_single_document_interface():
    """putInput / fetchResults / getStatistics / reset (patternMatcher.cpp:131-331)."""
    case = l2_cases.load("simple_token_pattern_match.json")
    m = spa.PatternMatcherInstance()
    l2_cases.build_simple(m, case)
    ctx = m.createContext()
    for lx in l2_cases.simple_doc(case):
        ctx.putInput(int(lx[0]), int(lx[1]), int(lx[3]), int(lx[4]))
    res, items = ctx.fetchResults()
    assert sorted({(m.patternName(int(r[0])), int(r[1])) for r in res}) == sorted(
        (r["name"], p) for r in case["rules"] for p in r["expected_ordpos"])
    st = ctx.getStatistics()
    assert st["nofProgramsInstalled"] == 18 and st["nofSignalsFired"] == 26
    ctx.reset()
    res, items = ctx.fetchResults()
    assert len(res) == 0


def test_failed_allocation_leaves_no_stale_capacity():
    """A hipMalloc that fails while the output buffers grow must not leave the old capacity next to a
    released buffer: the next launch reallocates (or reports an error), it never writes through NULL."""
    import os
    rules = synth.random_rules(300, 40, 91)
    lex, offs = synth.random_documents(20, 200, 40, 92)
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    synth.apply_rules(m, rules)
    synth.apply_rules(o, rules)
    ref = o.run(synth.lexems5(lex), offs)
    ctx = m.createContext()
    _compare(ctx.matchDocs(lex, offs), ref, 20)           # buffers exist, capacities > 0
    ctx.reserveOutput(12000000, 12000000)                  # the next launch has to grow both output buffers
    from struspattern_amd import capi
    capi.lib().sp_test_fail_alloc_above(64 << 20)
    try:
        with pytest.raises(spa.PatternError):
            ctx.matchDocs(lex, offs)
    finally:
        capi.lib().sp_test_fail_alloc_above(0)
    _compare(ctx.matchDocs(lex, offs), ref, 20)           # reallocated: same results as before
