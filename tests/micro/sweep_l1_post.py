# post-processing kernel of the lexer: waves per CU (SPA_L1_POST_WAVES_PER_CU) on the 10k-regex set; run once per
# build variant (tests/micro/ab.sh ... occ5:"-DSPA_L1_POST_WAVES_PER_EU=5" ...)
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
vocab = synth.vocabulary(30000, 1)
pats = synth.lexer_patterns(10000, vocab, 1)
text, offs = synth.text_documents(ndocs, 65536, vocab, 2)
lx = spa.PatternLexerInstance(); synth.apply_lexer_patterns(lx, pats)
ctx = lx.createContext()
d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
for per_cu in (20, 24, 28, 32):
    os.environ["SPA_L1_POST_WAVES_PER_CU"] = str(per_cu)
    best = None
    for it in range(5):
        ctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, len(text), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            ctx.reserveOutput(int(c["lexems"]*1.2)+1024); ctx.growArena(); continue
        a, b = ctx.lastKernelMsSplit()
        best = (a, b) if best is None or b < best[1] else best
    print("post waves/CU %d: scan %.1f ms post %.1f ms (%d docs, %.0f MB)" % (per_cu, best[0], best[1], ndocs, len(text)/1e6), flush=True)
