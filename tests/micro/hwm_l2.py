# where does the LDS tier overflow?  (PROF2 build, SPA_L2_TIER=ldsonly)
import sys, collections, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
nd = 2000
for op in (None, "sequence"):
    rules = synth.random_rules(10000, 10000, 2, op)
    lex, offs = synth.random_documents(nd, 1000, 10000, 1000)
    m = spa.PatternMatcherInstance(); synth.apply_rules(m, rules)
    ctx = m.createContext()
    d_lex = torch.from_numpy(lex.view(np.int32)).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    ctx.reserveOutput(40000000, 80000000)
    ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), nd, len(lex), 0)
    c = ctx.batchCounters()
    bf = ctx.batchFetch()
    st = np.asarray(bf.status); ds = np.asarray(bf.stats).reshape(-1, 4)
    bad = ds[st == 2]
    print("op", op, "failed", c["failed_docs"], "of", nd)
    print(" lines:", collections.Counter(int(x) for x in bad[:, 0]).most_common(8))
    if len(bad): print(" ruleUsed max/median", bad[:,1].max(), np.median(bad[:,1]), "trigUsed", bad[:,2].max(), np.median(bad[:,2]), "itemUsed", bad[:,3].max(), np.median(bad[:,3]))
