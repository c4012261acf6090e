import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
for op in (None, "sequence"):
    rules = synth.random_rules(10000, 10000, 2, op)
    lex, offs = synth.random_documents(6144, 1000, 10000, 1000)
    m = spa.PatternMatcherInstance(); synth.apply_rules(m, rules)
    ctx = m.createContext()
    d_lex = torch.from_numpy(lex.view(np.int32)).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    for it in range(14):
        ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), 6144, len(lex), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            st = ctx.batchStatus(6144); codes = set(int(x) for x in st[st!=0])
            if 9 in codes: ctx.reserveOutput(int(c["results"]*1.2)+1024, int(c["items"]*1.2)+1024)
            if 2 in codes: ctx.growArena()
            continue
        ms = ctx.lastKernelMs(); p = c["prof"]; tot = float(sum(p)) or 1.0
        print("op=%s: %.1f ms, %d events -> %.2f M ev/s; slots %.2f %.2f %.2f %.2f (ticks/event %.0f)" % (
            op, ms, c["events"], c["events"]/ms/1e3, p[0]/tot, p[1]/tot, p[2]/tot, p[3]/tot, tot/c["events"]), flush=True)
        break
