"""GPU parity of the fused pipeline: text -> lexer kernel -> (lexems stay in HBM) -> rule automaton
kernel, against oracle lexer -> oracle automaton on the same documents."""
import os

import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth

pytestmark = pytest.mark.gpu


def _device_pipeline(pats, rules, text, offs):
    """text -> lexer kernel -> lexems stay in HBM -> rule automaton kernel (the path bench.py times);
    returns (lexer counters, fetched matcher batch)."""
    import torch
    ndocs = len(offs) - 1
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    m = spa.PatternMatcherInstance()
    synth.apply_rules(m, rules)
    lctx, mctx = lx.createContext(), m.createContext()
    d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    lc = mc = None
    for _ in range(8):
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
    assert lc["failed_docs"] == 0 and mc is not None and mc["failed_docs"] == 0
    return lc, mctx.batchFetch()


def _oracle_pipeline(pats, rules, text, offs, nthreads=8):
    ol = oracle.L1Lexer()
    synth.apply_lexer_patterns(ol, pats)
    om = oracle.L2Matcher()
    synth.apply_rules(om, rules)
    lex, loffs = ol.matchDocs(text, offs, nthreads=nthreads)
    return lex, om.run(synth.lexems5(lex), loffs, nthreads=nthreads)


def _assert_same(gpu, ref):
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results[:, :7], ref.results[:, :7])
    assert np.array_equal(gpu.results[:, 8], ref.results[:, 8])
    assert np.array_equal(gpu.items, ref.items)
    assert np.array_equal(gpu.stats, ref.stats)


@pytest.mark.parametrize("npat,nrules,ndocs,docbytes,seed", [(200, 500, 24, 3000, 1), (600, 2000, 12, 4000, 2)])
def test_pipeline_parity(npat, nrules, ndocs, docbytes, seed):
    vocab = synth.vocabulary(2000, 5)
    pats, rules = synth.pipeline_workload(npat, nrules, vocab, seed)
    text, offs = synth.text_documents(ndocs, docbytes, vocab, 50 + seed)
    lc, gpu = _device_pipeline(pats, rules, text, offs)
    lex, ref = _oracle_pipeline(pats, rules, text, offs)
    assert len(lex) == lc["lexems"] and len(ref.results) > 0
    _assert_same(gpu, ref)


def test_headline_workload_parity():
    """BASELINE.json configs[4] as a whole: the exact tables bench.py builds (10 000 regexes + sentence
    delimiter, 10 000 token rules, vocabulary and seeds of bench.py) over UTF-8 documents of both
    document shapes the bench reports (16 KiB and the survey's 64 KiB): lexems, results in firing order,
    items and statistics of the fused device pipeline equal the oracle's."""
    ncores = min(16, len(os.sched_getaffinity(0)))
    vocab = synth.vocabulary(30000, 1)
    pats, rules = synth.pipeline_workload(10000, 10000, vocab, seed=4)
    t16, o16 = synth.text_documents(40, 16384, vocab, seed=1000, utf8=True)
    t64, o64 = synth.text_documents(6, 65536, vocab, seed=1001, utf8=True)
    text = t16 + t64
    offs = np.concatenate([o16, o64[1:] + o16[-1]]).astype(np.uint64)
    lc, gpu = _device_pipeline(pats, rules, text, offs)
    lex, ref = _oracle_pipeline(pats, rules, text, offs, nthreads=ncores)
    assert len(lex) == lc["lexems"] and len(ref.results) > len(lex)
    _assert_same(gpu, ref)


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
