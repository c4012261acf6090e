"""One-off wide parity sweep of the round-3 lexer kernels (lane-per-stream scan, words kernel, cluster handler): many seeds of the
synthetic lexer workload with word shapes, documents whole and cut into 1 KiB scan chunks, the handler by clusters and one report
after the other.  Not part of the test suite.  Usage: python tests/micro/sweep_parity_r3.py [nseeds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import oracle
import struspattern_amd as spa
from struspattern_amd import synth

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
bad = 0
t0 = time.time()
vocabs = {n: synth.vocabulary(n, 77) for n in (1500, 4000)}
for s in range(nseeds):
    npat = int(rng.choice([40, 64, 200, 700, 1500, 3000]))
    ndocs = int(rng.integers(3, 12))
    docbytes = int(rng.integers(200, 7000)) if s % 5 else int(rng.integers(20000, 70000))
    utf8 = bool(rng.integers(0, 2))
    vocab = vocabs[4000 if npat > 1000 else 1500]
    pats = synth.lexer_patterns(npat, vocab, 300 + s)
    text, offs = synth.text_documents(ndocs, docbytes, vocab, 900 + s, utf8=utf8)
    o = oracle.L1Lexer()
    synth.apply_lexer_patterns(o, pats)
    ref, roffs = o.matchDocs(text, offs, nthreads=8)
    for mode, env in (("plain", {}), ("chunks", {"SPA_L1_CHUNK_BYTES": "1024"}), ("sequential handler", {"SPA_L1_POST_SEQ": "1"})):
        for k in ("SPA_L1_CHUNK_BYTES", "SPA_L1_POST_SEQ"):
            os.environ.pop(k, None)
        os.environ.update(env)
        lx = spa.PatternLexerInstance()
        synth.apply_lexer_patterns(lx, pats)
        ctx = lx.createContext()
        gpu = ctx.matchDocs(text, offs)
        ok = (not gpu.status.any()) and np.array_equal(gpu.doc_offsets, roffs) and np.array_equal(gpu.lexems, ref)
        if not ok:
            bad += 1
            print("MISMATCH seed %d (%d patterns, %d docs x %d bytes, utf8 %s) mode %s" % (s, npat, ndocs, docbytes, utf8, mode), flush=True)
    if s % 4 == 3:
        print("%d seeds done (%.0f s, %d mismatches), last: %d patterns, %d lexems, scan kernel %s, words kernel %s" % (
            s + 1, time.time() - t0, bad, npat, len(ref), ctx.scanKernelName(), ctx.wordsKernelName()), flush=True)
print("SWEEP", "FAILED" if bad else "OK", nseeds, "seeds x 3 modes")
sys.exit(1 if bad else 0)
