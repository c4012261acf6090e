"""One-off wide parity sweep of the lexer: many more seeds of tests/test_l1_gpu.py's random regex sets and
synthetic workloads.  Usage: python tests/micro/sweep_parity_l1.py [nseeds]"""
import sys
import time

sys.path.insert(0, "/root/repo")
from tests import test_l1_gpu as t1

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
t0 = time.time()
for s in range(nseeds):
    try:
        t1.test_random_regex_sets(100 + s)      # seeds 5100.. (the suite uses 5000..5003)
    except AssertionError as e:
        bad += 1
        print("MISMATCH random_regex_sets seed", 100 + s, str(e)[:300], flush=True)
    if s % 5 == 4:
        print("%d seeds done (%.0f s, %d mismatches)" % (s + 1, time.time() - t0, bad), flush=True)
for k, (npat, utf8) in enumerate([(100, True), (300, False), (900, True), (2000, False), (5200, True), (7000, False)]):
    try:
        t1.test_synthetic_lexer_workload(npat, 10, 2500, utf8, 40 + k)
    except AssertionError as e:
        bad += 1
        print("MISMATCH synthetic", npat, utf8, str(e)[:300], flush=True)
# round 2: approximate literal tables; random regex sets with every document cut into 64- and 256-byte scan chunks
import os
for s in range(nseeds):
    try:
        t1.test_random_approximate_literal_tables(100 + s)
    except AssertionError as e:
        bad += 1
        print("MISMATCH approximate literal tables seed", 100 + s, str(e)[:300], flush=True)
print("approximate tables done (%.0f s, %d mismatches)" % (time.time() - t0, bad), flush=True)
for chunk in ("64", "256"):
    os.environ["SPA_L1_CHUNK_BYTES"] = chunk
    for s in range(nseeds):
        try:
            t1.test_random_regex_sets(300 + s)
        except AssertionError as e:
            bad += 1
            print("MISMATCH random_regex_sets chunk", chunk, "seed", 300 + s, str(e)[:300], flush=True)
    print("chunk %s done (%.0f s, %d mismatches)" % (chunk, time.time() - t0, bad), flush=True)
os.environ.pop("SPA_L1_CHUNK_BYTES", None)
print("SWEEP", "FAILED" if bad else "OK", nseeds, "seeds")
sys.exit(1 if bad else 0)
