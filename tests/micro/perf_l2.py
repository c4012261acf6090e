import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for op in (None, "sequence"):
    rules = synth.random_rules(10000, 10000, 2, op)
    lex, offs = synth.random_documents(nd, 1000, 10000, 1000)
    m = spa.PatternMatcherInstance(); synth.apply_rules(m, rules)
    ctx = m.createContext()
    d_lex = torch.from_numpy(lex.view(np.int32)).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    best = None
    for it in range(14):
        ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), nd, len(lex), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            st = ctx.batchStatus(nd); codes = set(int(x) for x in st[st!=0])
            if 9 in codes: ctx.reserveOutput(int(c["results"]*1.2)+1024, int(c["items"]*1.2)+1024)
            if 2 in codes: ctx.growArena()
            continue
        ms = ctx.lastKernelMs(); best = ms if best is None else min(best, ms)
    print("op=%s docs=%d: %.1f ms, %d events -> %.2f M ev/s, %.1f M matches/s" % (op, nd, best, c["events"], c["events"]/best/1e3, c["results"]/best/1e3), flush=True)
