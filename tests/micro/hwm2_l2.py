# high-water marks of the per-document state (PROF2 build, SPA_L2_TIER=global)
import sys, collections, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
nd = 2000
def report(name, m, lex, offs):
    ctx = m.createContext()
    d_lex = torch.from_numpy(lex.view(np.int32)).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    ctx.reserveOutput(40000000, 80000000)
    ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), len(offs)-1, len(lex), 0)
    c = ctx.batchCounters(); bf = ctx.batchFetch()
    ds = np.asarray(bf.stats).reshape(-1, 4)
    print(name, "failed", c["failed_docs"], "events/doc", c["events"]/(len(offs)-1))
    for i, nm in enumerate(("refs", "rules", "trigs", "items")):
        v = ds[:, i]
        print("  %-6s median %d  p90 %d  p99 %d  max %d" % (nm, np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
for op in (None, "sequence"):
    rules = synth.random_rules(10000, 10000, 2, op)
    lex, offs = synth.random_documents(nd, 1000, 10000, 1000)
    m = spa.PatternMatcherInstance(); synth.apply_rules(m, rules)
    report("l2 op=%s" % op, m, lex, offs)
