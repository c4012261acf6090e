import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import struspattern_amd as spa
from struspattern_amd import synth
vocab = synth.vocabulary(30000, 1)
for npat, ndocs in ((10000, int(sys.argv[1]) if len(sys.argv) > 1 else 4096),):
    pats = synth.lexer_patterns(npat, vocab, 1)
    t0=time.time(); text, offs = synth.text_documents(ndocs, 65536, vocab, 2); tg=time.time()-t0
    lx = spa.PatternLexerInstance(); t0=time.time(); synth.apply_lexer_patterns(lx, pats); tc=time.time()-t0
    T = lx.dumpTables()
    ctx = lx.createContext()
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    for it in range(4):
        ctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, len(text), 0)
        c = ctx.batchCounters()
        if c["failed_docs"]:
            st = ctx.batchStatus(ndocs); print("failed", sorted(set(int(x) for x in st[st!=0])))
            ctx.reserveOutput(int(c["lexems"]*1.2)+1024); ctx.growArena(); continue
        ms = ctx.lastKernelMs()
        pr = c["prof"]; tot = pr[0] or 1
        if __import__("os").environ.get("SPA_SHOW_WORDS_PROF"): print("   words kernel prof (cycles): runs %d literal %d shapes+merge %d walks %d" % tuple(pr))
        elif pr[0] > 10**7: print("   post kernel prof: cycles/lexem %.0f; shares literals %.2f starts %.2f handler %.2f rest(merge+emit) %.2f" % (pr[0]/max(1,c["lexems"]), pr[1]/tot, pr[2]/tot, pr[3]/tot, (pr[0]-pr[1]-pr[2]-pr[3])/tot))
        print("   raw reports %d, %d documents scanned again; scan %.2f ms words %.2f ms post %.2f ms" % ((c["raw_reports"], c.get("rescanned_docs", -1)) + ctx.lastKernelMsSplit3())); print("npat %d passes %d classes %d maxEx %d: %d docs %.1f MB: kernel %.1f ms -> %.3f GB/s, %d lexems (gen %.1fs compile %.1fs)" % (
            npat, int(T[0]), int(T[1]), int(T[2]), ndocs, len(text)/1e6, ms, len(text)/ms/1e6, c["lexems"], tg, tc), flush=True)
