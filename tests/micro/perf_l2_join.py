# the join prototype (SPA_L2_JOIN=1, csrc/l2_join.h) beside the exact engine on the sequence-only automaton workload:
# 10k two-term rules (sequence only, or the 5-operator mix: `mix` as second argument) x 1000-token documents (exact engine:
# optimized automaton as in bench.py; its result count may differ by the optimizer's alternative keys, ~0.02 %)
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
op = None if len(sys.argv) > 2 and sys.argv[2] == "mix" else "sequence"
rules = synth.random_rules(10000, 10000, 2, op)
lex, offs = synth.random_documents(nd, 1000, 10000, 1000)
d_lex = torch.from_numpy(lex.view(np.int32)).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
for join in ("0", "1"):
    os.environ["SPA_L2_JOIN"] = join
    m = spa.PatternMatcherInstance(); synth.apply_rules(m, rules)
    ctx = m.createContext()
    best = None
    for it in range(14):
        ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), nd, len(lex), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            st = ctx.batchStatus(nd); codes = set(int(x) for x in st[st!=0])
            if 9 in codes: ctx.reserveOutput(int(c["results"]*1.2)+1024, int(c["items"]*1.2)+1024)
            if 2 in codes: ctx.growArena()
            if not (codes & {2, 9}): print("failed", codes); break
            continue
        ms = ctx.lastKernelMs(); best = ms if best is None else min(best, ms)
    print("kernel kind %d (%s): %.1f ms, %d events -> %.1f M events/s, %d results (%.1f M matches/s)" % (
        ctx.kernelKind(), {0: "general", 1: "LDS-resident exact engine", 2: "join prototype"}[ctx.kernelKind()], best, c["events"], c["events"]/best/1e3, c["results"], c["results"]/best/1e3), flush=True)
