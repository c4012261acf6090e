"""GPU parity of the fast tier of the rule automaton (flat rule sets, document state in LDS:
struspattern_amd/csrc/l2_fast_kernel.hip) against the CPU oracle, bit for bit: results in firing order,
captured items, statistics.  Every case runs in four configurations of the same library:
the default capacities, LDS capacities so small that rules and bucket chunks live in the spill area,
a rule capacity so small that documents are handed over to the general kernel (list mode), and the
general kernel alone -- all four must give the oracle's output."""
import os

import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth

pytestmark = pytest.mark.gpu

CONFIGS = {
    "default": {},
    "spill": {"SPA_L2_FAST_SIZE": "t"},
    "handover": {"SPA_L2_FAST_SIZE": "t", "SPA_L2_FAST_MAXRULES": "24", "SPA_L2_FAST_MAXSTAGED": "40"},
    "general": {"SPA_L2_FAST": "0"},
}


def _context(m, config):
    env = CONFIGS[config]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return m.createContext()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _check(build, lex4, offs, config, origseg=None, expect_fast=True, handover=None):
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    build(m)
    build(o)
    assert m.fastTier()[0] == expect_fast, m.fastTier()
    ctx = _context(m, config)
    gpu = ctx.matchDocs(lex4, offs, origseg)
    handed = ctx.batchCounters()["handed_over"]
    if config == "handover" and expect_fast:
        assert handed > 0
    if config == "general" or handover is False:
        assert handed == 0
    l5 = synth.lexems5(lex4)
    if origseg is not None:
        l5[:, 2] = origseg
    ref = o.run(l5, offs)
    ndocs = len(offs) - 1
    assert np.array_equal(gpu.status, np.zeros(ndocs, np.int32))
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results[:, :7], ref.results[:, :7])
    assert np.array_equal(gpu.stats, ref.stats)
    assert np.array_equal(gpu.results[:, 8], ref.results[:, 8])
    assert np.array_equal(gpu.items, ref.items)
    return ref


@pytest.mark.parametrize("config", list(CONFIGS))
@pytest.mark.parametrize("nrules,nfeat,ndocs,docsize,seed,op,optimize", [
    (2000, 200, 200, 300, 121, None, True),
    (2000, 200, 200, 300, 122, None, False),
    (3000, 600, 100, 800, 123, "sequence", True),
    (800, 40, 80, 400, 124, "within", True),
    (800, 40, 80, 400, 125, "sequence_struct", True),
    (800, 40, 80, 400, 126, "within_struct", True),
    (400, 25, 80, 300, 127, "any", True),
])
def test_two_term_rules(config, nrules, nfeat, ndocs, docsize, seed, op, optimize):
    rules = synth.random_rules(nrules, nfeat, seed, op)
    lex, offs = synth.random_documents(ndocs, docsize, nfeat, seed + 1000)
    ref = _check(lambda x: synth.apply_rules(x, rules, compile=optimize), lex, offs, config, handover=(None if config == "handover" else False))
    assert len(ref.results) > 0


def _apply_mixed(m, rules, compile_):
    for name, op, rg, card, params, variables in rules:
        n = len(params)
        if op in ("sequence_struct", "within_struct"):
            m.pushTerm(synth.DELIM)
            n += 1
        for t, v in zip(params, variables):
            m.pushTerm(t)
            if v:
                m.attachVariable(v)
        m.pushExpression(op, n, rg, card)
        m.definePattern(name, "", not name.startswith("_"))
    if compile_:
        m.compile()


@pytest.mark.parametrize("config", list(CONFIGS))
@pytest.mark.parametrize("seed,compile_", [(131, True), (132, False), (133, True)])
def test_flat_rules_of_every_shape(config, seed, compile_):
    """1-3 terms, all flat operators incl. sequence_imm, cardinalities that let `any` take more than one event,
    repeated terms (several key triggers of one program), terms with and without variables, invisible
    patterns, ordinal positions with gaps and repeats, position 0, several segments."""
    rng = np.random.default_rng(seed)
    nfeat = 14
    ops = ["sequence", "sequence_imm", "within", "any", "sequence_struct", "within_struct"]
    rules = []
    for ni in range(500):
        op = ops[int(rng.integers(0, len(ops)))]
        struct = op.endswith("_struct")
        nterms = int(rng.integers(1 if not struct else 1, 3 if struct else 4))
        if rng.random() < 0.15:
            t = int(rng.integers(1, nfeat + 1))
            params = [t] * nterms                     # the same term several times
        else:
            params = [int(x) for x in rng.integers(1, nfeat + 1, size=nterms)]
        if rng.random() < 0.05:
            params[0] = synth.DELIM
        variables = [("v%d" % int(rng.integers(0, 5))) if rng.random() < 0.6 else None for _ in params]
        card = 0
        if op == "any" and rng.random() < 0.5:
            card = int(rng.integers(1, 3))
        rg = int(rng.integers(0, 64))
        name = ("_" if rng.random() < 0.1 else "") + "r%d" % ni
        rules.append((name, op, rg, card, params, variables))
    ndocs, n = 48, 350
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    seg = np.zeros(ndocs * n, np.uint32)
    for d in range(ndocs):
        steps = rng.choice([0, 0, 1, 1, 1, 1, 2, 3, 7, 40, 63, 64, 65, 200], size=n)
        steps[0] = 0 if d % 3 == 0 else 1           # some documents start at ordinal position 0
        ids = rng.integers(1, nfeat + 1, size=n)
        ids[rng.random(n) < 0.07] = synth.DELIM
        lex[d * n:(d + 1) * n, 0] = ids
        lex[d * n:(d + 1) * n, 1] = np.cumsum(steps)
        lex[d * n:(d + 1) * n, 2] = np.arange(n) * 3
        lex[d * n:(d + 1) * n, 3] = 2
        seg[d * n:(d + 1) * n] = np.arange(n) // 120
    ref = _check(lambda x: _apply_mixed(x, rules, compile_), lex, offs, config, origseg=seg)
    assert len(ref.results) > 1000


@pytest.mark.parametrize("config", ["default", "spill"])
def test_frequent_key_event_installs_hundreds_of_programs(config):
    """one term keys 700 programs (several 64-program batches per event, bursts far beyond the LDS capacities)"""
    rng = np.random.default_rng(141)
    rules = []
    for ni in range(700):
        op = ["sequence", "within", "sequence_struct", "any"][ni % 4]
        rules.append(("k%d" % ni, op, int(rng.integers(1, 9)), [1, int(rng.integers(2, 30))]))
    for ni in range(300):
        rules.append(("o%d" % ni, "sequence", int(rng.integers(1, 12)), [int(rng.integers(2, 30)), int(rng.integers(1, 30))]))
    lex, offs = synth.random_documents(40, 200, 29, 142)
    ref = _check(lambda x: synth.apply_rules(x, rules), lex, offs, config)
    assert ref.stats[:, 0].max() > 20000


def test_rule_sets_outside_the_fast_tier_still_match():
    """nested rule sets take the general kernel as before (the choice is not observable)"""
    def build(m):
        m.pushTerm(1)
        m.pushTerm(2)
        m.attachVariable("b")
        m.pushExpression("sequence", 2, 3, 0)
        m.pushTerm(3)
        m.pushExpression("within", 2, 8, 0)
        m.definePattern("outer", "", True)
        m.compile()
    lex, offs = synth.random_documents(30, 200, 4, 151)
    _check(build, lex, offs, "default", expect_fast=False)
