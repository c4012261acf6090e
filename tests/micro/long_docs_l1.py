# lexer on a batch of few long documents: scan in chunks (default, 32 KiB) vs one wave per document (SPA_L1_CHUNK_BYTES = 1 GiB)
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
docbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
vocab = synth.vocabulary(30000, 1)
pats = synth.lexer_patterns(10000, vocab, 1)
text, offs = synth.text_documents(ndocs, docbytes, vocab, 2)
lx = spa.PatternLexerInstance(); synth.apply_lexer_patterns(lx, pats)
ctx = lx.createContext()
d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
for chunk in ("1073741824", None):
    if chunk: os.environ["SPA_L1_CHUNK_BYTES"] = chunk
    else: os.environ.pop("SPA_L1_CHUNK_BYTES", None)
    best = None
    for it in range(8):
        ctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, len(text), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            ctx.reserveOutput(int(c["lexems"]*1.2)+1024); ctx.growArena(); continue
        a, w_, b = ctx.lastKernelMsSplit3()
        best = (a, w_, b) if best is None or a + w_ + b < sum(best) else best
    print("%s: %d docs x %.1f MB: scan %.1f ms words %.1f ms post %.1f ms, %d units, %d documents scanned again, %d lexems" % (
        "one wave per document" if chunk else "32 KiB chunks", ndocs, docbytes/1e6, best[0], best[1], best[2], c["scan_units"], c["rescanned_docs"], c["lexems"]), flush=True)
