"""GPU parity of the fused pipeline: text -> lexer kernel -> (lexems stay in HBM) -> rule automaton
kernel, against oracle lexer -> oracle automaton on the same documents."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("npat,nrules,ndocs,docbytes,seed", [(200, 500, 24, 3000, 1), (600, 2000, 12, 4000, 2)])
def test_pipeline_parity(npat, nrules, ndocs, docbytes, seed):
    import torch
    vocab = synth.vocabulary(2000, 5)
    pats, rules = synth.pipeline_workload(npat, nrules, vocab, seed)
    text, offs = synth.text_documents(ndocs, docbytes, vocab, 50 + seed)
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    m = spa.PatternMatcherInstance()
    synth.apply_rules(m, rules)
    lctx, mctx = lx.createContext(), m.createContext()
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        lo = lctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, len(text), stream)
        lc = lctx.batchCounters()
        if lc["failed_docs"]:
            st = lctx.batchStatus(ndocs)
            assert set(int(x) for x in st[st != 0]) <= {2, 9}
            lctx.reserveOutput(int(lc["lexems"] * 1.2) + 1024)
            lctx.growArena()
            continue
        mctx.matchLexedDevice(lo.d_lexems, lo.d_doc_ranges, ndocs, int(lc["lexems"]), stream)
        mc = mctx.batchCounters()
        if mc["failed_docs"] == 0:
            break
        mctx.reserveOutput(int(mc["results"] * 1.2) + 1024, int(mc["items"] * 1.2) + 1024)
        mctx.growArena()
    assert lc["failed_docs"] == 0 and mc["failed_docs"] == 0
    gpu = mctx.batchFetch()

    ol = oracle.L1Lexer()
    synth.apply_lexer_patterns(ol, pats)
    om = oracle.L2Matcher()
    synth.apply_rules(om, rules)
    lex, loffs = ol.matchDocs(text, offs, nthreads=8)
    ref = om.run(synth.lexems5(lex), loffs)
    assert len(lex) == lc["lexems"] and len(ref.results) > 0
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results[:, :7], ref.results[:, :7])
    assert np.array_equal(gpu.items, ref.items)
    assert np.array_equal(gpu.stats, ref.stats)
