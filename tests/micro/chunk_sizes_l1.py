import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
ndocs=int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vocab = synth.vocabulary(30000, 1)
pats = synth.lexer_patterns(10000, vocab, 1)
text, offs = synth.text_documents(ndocs, 65536, vocab, 2)
lx = spa.PatternLexerInstance(); synth.apply_lexer_patterns(lx, pats)
ctx = lx.createContext()
d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
for chunk in (None, "32768", "16384", "4096"):
    if chunk: os.environ["SPA_L1_CHUNK_BYTES"] = chunk
    else: os.environ.pop("SPA_L1_CHUNK_BYTES", None)
    best=None
    for it in range(6):
        ctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, len(text), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            ctx.reserveOutput(int(c["lexems"]*1.2)+1024); ctx.growArena(); continue
        a,b = ctx.lastKernelMsSplit(); best=(a,b) if best is None or a<best[0] else best
    print("chunk", chunk, "scan %.1f post %.1f units %d rescanned %d" % (best[0], best[1], c["scan_units"], c["rescanned_docs"]), flush=True)
