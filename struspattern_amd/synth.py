"""Synthetic workload generators (fixed seeds) restating the reference's test generators:
tests/utils/testUtils.cpp:39-96 (Zipf, random documents) and
tests/randomTokenPatternMatch/src/testRandomTokenPatternMatch.cpp:49-123 (random two-term rules).
The reference seeds rand() from the calendar date; these use numpy's PCG64 with explicit seeds."""
import numpy as np

DELIM = 1 << 24  # termId(SentenceDelim, 0), tests/utils/testUtils.cpp:74-77
OPS5 = ["sequence", "within", "sequence_struct", "within_struct", "any"]  # testRandomTokenPatternMatch.cpp:108-115


def zipf_cum(n, S=0.0):
    """ZipfDistribution ctor (testUtils.cpp:39-54): c[0]=1, c[i]=c[i-1]+1/(i+1)^S; S==0 means exponent 1."""
    k = np.arange(1, n + 1, dtype=np.float64)
    w = 1.0 / (k ** S) if S > np.finfo(np.float64).eps else 1.0 / k
    w[0] = 1.0
    return np.cumsum(w)


def zipf_sample(cum, rng, size=None):
    """ZipfDistribution::random (testUtils.cpp:56-72): first i with v < c[i], returned 1-based."""
    v = cum[-1] * rng.random(size)
    return (np.searchsorted(cum, v, side="right") + 1).astype(np.uint32)


def random_documents(ndocs, docsize, nfeatures, seed):
    """createRandomDocument (testUtils.cpp:79-96) for a whole collection.
    Returns lexems (n,4) u32 [id, ordpos, origpos=item index, origsize=1] and doc_offsets (ndocs+1,) u64."""
    rng = np.random.default_rng(seed)
    cum = zipf_cum(nfeatures)
    tok = zipf_sample(cum, rng, (ndocs, docsize))
    has_delim = rng.integers(0, 12, (ndocs, docsize)) == 0
    per_tok = 1 + has_delim.astype(np.int64)
    doc_len = per_tok.sum(axis=1)
    doc_offsets = np.zeros(ndocs + 1, np.uint64)
    doc_offsets[1:] = np.cumsum(doc_len)
    n = int(doc_offsets[-1])
    # position of each token item inside the flat array
    flat_per = per_tok.reshape(-1)
    tok_idx = np.cumsum(flat_per) - flat_per          # index of the token item
    lex = np.zeros((n, 4), np.uint32)
    pos = np.tile(np.arange(1, docsize + 1, dtype=np.uint32), ndocs)
    lex[tok_idx, 0] = tok.reshape(-1)
    lex[tok_idx, 1] = pos
    didx = tok_idx[has_delim.reshape(-1)] + 1
    lex[didx, 0] = DELIM
    lex[didx, 1] = pos[has_delim.reshape(-1)]
    # origpos = index of the item within its document, origsize = 1 (testRandomTokenPatternMatch.cpp:145)
    doc_of_item = np.repeat(np.arange(ndocs), doc_len)
    lex[:, 2] = (np.arange(n, dtype=np.uint64) - doc_offsets[doc_of_item]).astype(np.uint32)
    lex[:, 3] = 1
    return lex, doc_offsets


def random_rules(nrules, nfeatures, seed, op=None):
    """createRules (testRandomTokenPatternMatch.cpp:90-123) as data: list of (name, op, range, [t0, t1]).
    Every rule gets its own pattern name so matches are attributable."""
    rng = np.random.default_rng(seed)
    featcum = zipf_cum(nfeatures, 0.8)
    rangecum = zipf_cum(10, 1.7)
    opcum = zipf_cum(5)
    rules = []
    for ni in range(nrules):
        rg = int(zipf_sample(rangecum, rng)) + 1
        p0 = int(zipf_sample(featcum, rng))
        p1 = int(zipf_sample(featcum, rng))
        o = op if op else OPS5[int(zipf_sample(opcum, rng)) - 1]
        rules.append(("%s_%d" % (o, ni), o, rg, [p0, p1]))
    return rules


def apply_rules(m, rules, compile=True):
    """createTermOpRule/createTermOpPattern (testRandomTokenPatternMatch.cpp:49-88) on any object with the
    PatternMatcherInstanceInterface method names."""
    for name, o, rg, params in rules:
        n = len(params)
        if o in ("sequence_struct", "within_struct"):
            m.pushTerm(DELIM)
            n += 1
        for pi, t in enumerate(params):
            m.pushTerm(t)
            m.attachVariable("A%d" % pi)
        m.pushExpression(o, n, rg, 0)
        m.definePattern(name, "", True)
    if compile:
        m.compile()


def lexems5(lex4):
    """(n,4) [id, ordpos, origpos, origsize] -> (n,5) [id, ordpos, origseg=0, origpos, origsize]."""
    out = np.zeros((len(lex4), 5), np.uint32)
    out[:, 0] = lex4[:, 0]
    out[:, 1] = lex4[:, 1]
    out[:, 3] = lex4[:, 2]
    out[:, 4] = lex4[:, 3]
    return out
