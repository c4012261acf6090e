# phase shares of the automaton kernel on the pipeline workload (PROF=1 build)
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
vocab = synth.vocabulary(30000, 1)
pats, rules = synth.pipeline_workload(10000, 10000, vocab, seed=4)
text, offs = synth.text_documents(nd, 65536, vocab, seed=1000, utf8=True)
lxi = spa.PatternLexerInstance(); synth.apply_lexer_patterns(lxi, pats); lctx = lxi.createContext()
d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda(); d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
for it in range(6):
    o = lctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), nd, len(text), 0)
    c = lctx.batchCounters()
    if not c["failed_docs"]: break
    lctx.reserveOutput(int(c["lexems"]*1.2)+1024); lctx.growArena()
nlex = int(c["lexems"]); print("lexems", nlex, "L1 ms", lctx.lastKernelMs())
mi = spa.PatternMatcherInstance(); synth.apply_rules(mi, rules)
import os
for size in (sys.argv[2].split(",") if len(sys.argv) > 2 else [os.environ.get("SPA_L2_FAST_SIZE", "m")]):
    os.environ["SPA_L2_FAST_SIZE"] = size
    mctx = mi.createContext()
    best = None
    for it in range(16):
        mctx.matchLexedDevice(o.d_lexems, o.d_doc_ranges, nd, nlex, 0)
        c = mctx.batchCounters()
        if c["failed_docs"]:
            st = mctx.batchStatus(nd); codes = set(int(x) for x in st[st!=0])
            if 9 in codes: mctx.reserveOutput(int(c["results"]*1.2)+1024, int(c["items"]*1.2)+1024)
            if 2 in codes: mctx.growArena()
            continue
        ms = mctx.lastKernelMs()
        best = ms if best is None or ms < best else best
        if it >= 12: break
    print("kernel kind %d;" % mctx.kernelKind(), "pipeline L2 size %s: %.1f ms, %d docs, %d events (%.1f M ev/s), %d results, handed over %d" % (
        size, best, nd, c["events"], c["events"]/best/1e3, c["results"], c["handed_over"]), flush=True)
