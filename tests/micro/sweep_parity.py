"""One-off wide parity sweep of the automaton (many seeds of the generators of tests/test_l2_gpu.py and
tests/test_formats.py); not part of the test suite.  Usage: python tests/micro/sweep_parity.py [nseeds]"""
import random
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import oracle
import struspattern_amd as spa
from struspattern_amd import synth
from tests import test_formats as tf
from tests import test_l2_gpu as t2

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
t0 = time.time()


def check(what, gpu, ref, ndocs):
    global bad
    try:
        t2._compare(gpu, ref, ndocs)
        if getattr(ref, "result_format", None) is not None:
            assert np.array_equal(gpu.result_format, ref.result_format) and np.array_equal(gpu.item_format, ref.item_format)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", what, str(e)[:200], flush=True)


for s in range(nseeds):
    seed = 7000 + s
    rng = np.random.default_rng(seed)
    # token rules, all operators, with and without the optimizer
    for op in (None, "sequence", "within", "sequence_struct", "within_struct", "any"):
        nfeat = int(rng.integers(8, 120))
        rules = synth.random_rules(int(rng.integers(100, 1500)), nfeat, seed, op)
        lex, offs = synth.random_documents(40, int(rng.integers(50, 400)), nfeat, seed + 1)
        opt = bool(rng.integers(0, 2))
        gpu, ref, m, o = t2._run_both(lambda x: synth.apply_rules(x, rules, compile=opt), lex, offs)
        check("rules seed=%d op=%s opt=%s" % (seed, op, opt), gpu, ref, 40)
    # expression trees
    for maxdepth in (2, 4, 6):
        nfeat = int(rng.integers(4, 9))
        trees = [t2._random_tree(rng, nfeat, 0, maxdepth)[1] for _ in range(40)]

        def build(m):
            for i, push in enumerate(trees):
                push(m)
                m.definePattern("tree_%d" % i, "", True)
            m.compile()
        ndocs, n = 12, 200
        lex = np.zeros((ndocs * n, 4), np.uint32)
        offs = np.arange(ndocs + 1, dtype=np.uint64) * n
        for d in range(ndocs):
            ids = rng.integers(1, nfeat + 1, size=n)
            ids[rng.random(n) < 0.05] = synth.DELIM
            lex[d * n:(d + 1) * n, 0] = ids
            lex[d * n:(d + 1) * n, 1] = np.cumsum(rng.choice([1, 1, 1, 2], size=n))
            lex[d * n:(d + 1) * n, 2] = np.arange(n) * 2
            lex[d * n:(d + 1) * n, 3] = 1
        gpu, ref, m, o = t2._run_both(build, lex, offs)
        check("trees seed=%d depth=%d" % (seed, maxdepth), gpu, ref, ndocs)
    # programs with format strings
    prng = random.Random(seed)
    calls = tf._random_program(prng, 6)
    mt, omt = spa.PatternMatcherInstance(), oracle.L2Matcher()
    tf._apply(mt, calls)
    tf._apply(omt, calls)
    lex, offs = synth.random_documents(30, 150, 6, seed=seed + 2)
    gpu = mt.createContext().matchDocs(lex, offs)
    ref = omt.run(synth.lexems5(lex), offs, nthreads=4)
    check("formats seed=%d" % seed, gpu, ref, 30)
    print("seed %d done (%.0f s, %d mismatches so far)" % (seed, time.time() - t0, bad), flush=True)
print("SWEEP", "FAILED" if bad else "OK", nseeds, "seeds")
sys.exit(1 if bad else 0)
