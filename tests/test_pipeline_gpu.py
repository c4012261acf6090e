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


def test_one_long_document_among_short_ones():
    """Ragged batch: a 1.5 MB document (~250k lexems, every per-document capacity of both kernels has to
    grow several times) between short and empty ones; lexems, results, items and statistics of every
    document equal the oracle."""
    vocab = synth.vocabulary(3000, 77)
    pats, rules = synth.pipeline_workload(300, 800, vocab, seed=9)
    long_text, _ = synth.text_documents(1, 1500000, vocab, 901, utf8=True)
    short_text, short_offs = synth.text_documents(6, 2000, vocab, 902, utf8=True)
    docs = [short_text[int(short_offs[i]):int(short_offs[i + 1])] for i in range(6)]
    docs = docs[:2] + [b""] + [bytes(long_text)] + docs[2:] + [b""]
    text = b"".join(docs)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    lx, olx = spa.PatternLexerInstance(), oracle.L1Lexer()
    synth.apply_lexer_patterns(lx, pats)
    synth.apply_lexer_patterns(olx, pats)
    lb = lx.createContext().matchDocs(text, offs)
    ref_lex, ref_offs = olx.matchDocs(text, offs, nthreads=8)
    assert np.array_equal(lb.status, np.zeros(len(docs), np.int32))
    assert np.array_equal(lb.doc_offsets, ref_offs)
    assert np.array_equal(lb.lexems, ref_lex)
    assert lb.doc_offsets[4] - lb.doc_offsets[3] > 100000   # the long document
    mt, omt = spa.PatternMatcherInstance(), oracle.L2Matcher()
    synth.apply_rules(mt, rules)
    synth.apply_rules(omt, rules)
    mb = mt.createContext().matchDocs(lb.lexems, lb.doc_offsets)
    ref = omt.run(synth.lexems5(ref_lex), ref_offs, nthreads=8)
    assert np.array_equal(mb.status, np.zeros(len(docs), np.int32))
    assert np.array_equal(mb.doc_offsets, ref.doc_offsets)
    assert np.array_equal(mb.results[:, :7], ref.results[:, :7])
    assert np.array_equal(mb.results[:, 8], ref.results[:, 8])
    assert np.array_equal(mb.items, ref.items)
    assert np.array_equal(mb.stats, ref.stats)
