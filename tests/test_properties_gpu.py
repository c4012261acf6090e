"""Size-independent properties of the GPU path at sizes where the oracle would take too long
(BASELINE.json configs[1..3] shapes, scaled to what the GPU test run can afford):
 * batching invariance: a document's output does not depend on which other documents share the
   launch, on their order, or on how the batch is split (documents are independent units);
 * idempotence: running the same batch twice gives byte-identical output;
 * spot parity: a random sample of the documents equals the oracle bit for bit;
 * the `exclusive` option (src/patternMatcher.cpp:192-246) against the oracle."""
import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth

pytestmark = pytest.mark.gpu


def _per_doc(batch, i):
    r = batch.doc(i)
    items = [batch.items[x[7]:x[7] + x[8]].tolist() for x in r]
    return r[:, :7].tolist(), items


def test_l2_batching_invariance_and_sample_parity():
    ndocs = 3000
    rules = synth.random_rules(10000, 10000, 2)
    lex, offs = synth.random_documents(ndocs, 1000, 10000, 77)
    m = spa.PatternMatcherInstance()
    synth.apply_rules(m, rules)
    ctx = m.createContext()
    full = ctx.matchDocs(lex, offs)
    again = ctx.matchDocs(lex, offs)
    assert np.array_equal(full.results, again.results) and np.array_equal(full.items, again.items)
    assert np.array_equal(full.stats, again.stats)
    # reversed document order + split in two launches
    rng = np.random.default_rng(5)
    perm = rng.permutation(ndocs)
    plex = np.concatenate([lex[int(offs[d]):int(offs[d + 1])] for d in perm])
    poffs = np.concatenate([[0], np.cumsum([int(offs[d + 1] - offs[d]) for d in perm])]).astype(np.uint64)
    half = ndocs // 2
    a = ctx.matchDocs(plex[:int(poffs[half])], poffs[:half + 1])
    b = ctx.matchDocs(plex[int(poffs[half]):], poffs[half:] - poffs[half])
    for k in rng.choice(ndocs, size=200, replace=False):
        d = int(perm[k])
        got = _per_doc(a, int(k)) if k < half else _per_doc(b, int(k - half))
        assert got == _per_doc(full, d)
    # spot parity against the oracle
    sample = sorted(int(x) for x in rng.choice(ndocs, size=40, replace=False))
    o = oracle.L2Matcher()
    synth.apply_rules(o, rules)
    slex = np.concatenate([lex[int(offs[d]):int(offs[d + 1])] for d in sample])
    soffs = np.concatenate([[0], np.cumsum([int(offs[d + 1] - offs[d]) for d in sample])]).astype(np.uint64)
    ref = o.run(synth.lexems5(slex), soffs)
    for k, d in enumerate(sample):
        r = ref.results[int(ref.doc_offsets[k]):int(ref.doc_offsets[k + 1])]
        items = [ref.items[x[7]:x[7] + x[8]].tolist() for x in r]
        assert (r[:, :7].tolist(), items) == _per_doc(full, d)
        assert np.array_equal(full.stats[d], ref.stats[k])


def test_l1_batching_invariance_and_sample_parity():
    vocab = synth.vocabulary(30000, 1)
    pats = synth.lexer_patterns(256, vocab, 1)
    ndocs = 600
    text, offs = synth.text_documents(ndocs, 16384, vocab, 9)
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    ctx = lx.createContext()
    full = ctx.matchDocs(text, offs)
    again = ctx.matchDocs(text, offs)
    assert np.array_equal(full.lexems, again.lexems) and np.array_equal(full.doc_offsets, again.doc_offsets)
    rng = np.random.default_rng(6)
    perm = rng.permutation(ndocs)
    ptext = b"".join(text[int(offs[d]):int(offs[d + 1])] for d in perm)
    poffs = np.concatenate([[0], np.cumsum([int(offs[d + 1] - offs[d]) for d in perm])]).astype(np.uint64)
    shuffled = ctx.matchDocs(ptext, poffs)
    for k in range(ndocs):
        assert np.array_equal(shuffled.doc(k), full.doc(int(perm[k])))
    # the ordinal positions of a document restart at 1 and never decrease; origpos never decreases
    for d in range(0, ndocs, 37):
        lexd = full.doc(d)
        if len(lexd):
            assert lexd[0, 1] == 1 and np.all(np.diff(lexd[:, 1].astype(np.int64)) >= 0) and np.all(np.diff(lexd[:, 2].astype(np.int64)) >= 0)
    sample = sorted(int(x) for x in rng.choice(ndocs, size=8, replace=False))
    o = oracle.L1Lexer()
    synth.apply_lexer_patterns(o, pats)
    stext = b"".join(text[int(offs[d]):int(offs[d + 1])] for d in sample)
    soffs = np.concatenate([[0], np.cumsum([int(offs[d + 1] - offs[d]) for d in sample])]).astype(np.uint64)
    ref, roffs = o.matchDocs(stext, soffs, nthreads=8)
    for k, d in enumerate(sample):
        assert np.array_equal(ref[int(roffs[k]):int(roffs[k + 1])], full.doc(d))


def test_exclusive_option():
    rules = synth.random_rules(300, 20, 3)
    lex, offs = synth.random_documents(60, 200, 20, 4)

    def build(x):
        x.defineOption("exclusive")
        x.defineOption("maxResultSize", 30)
        synth.apply_rules(x, rules)
    m = spa.PatternMatcherInstance()
    o = oracle.L2Matcher()
    build(m)
    build(o)
    gpu = m.createContext().matchDocs(lex, offs)
    ref = o.run(synth.lexems5(lex), offs)
    assert len(ref.results) > 0
    assert np.array_equal(gpu.doc_offsets, ref.doc_offsets)
    assert np.array_equal(gpu.results[:, :7], ref.results[:, :7])
    assert np.array_equal(gpu.items, ref.items)
