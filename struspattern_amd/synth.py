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


# ---------------------------------------------------------------- level 1 synthetic workload
_LETTERS = "etaoinshrdlcumwfgypbvkjxqz"


def vocabulary(nwords, seed):
    """Synthetic vocabulary: distinct lower-case words of length 2..12 (letter frequencies roughly English)."""
    rng = np.random.default_rng(seed)
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    p = p / p.sum()
    words, seen = [], set()
    while len(words) < nwords:
        ln = int(rng.integers(2, 13))
        w = "".join(_LETTERS[i] for i in rng.choice(26, size=ln, p=p))
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def text_documents(ndocs, docbytes, vocab, seed, utf8=False):
    """Documents of about `docbytes` bytes: Zipf(s=1) words, separators ' ' / '. ' / '\\n', 2 % digit tokens,
    some capitalised words; with utf8=True 10 % of the words carry a 2-byte and 2 % a 3-byte code point.
    Returns (bytes of all documents, doc_offsets u64)."""
    rng = np.random.default_rng(seed)
    cum = zipf_cum(len(vocab), 1.0)
    parts = []
    offs = [0]
    total = 0
    for d in range(ndocs):
        nw = docbytes // 5 + 8
        idx = zipf_sample(cum, rng, nw) - 1
        r = rng.random(nw)
        sep = rng.random(nw)
        toks = []
        size = 0
        for k in range(nw):
            if r[k] < 0.02:
                w = str(int(rng.integers(0, 100000)))
                if rng.random() < 0.3:
                    w = w[:2] + "'" + "%03d" % int(rng.integers(0, 1000))
            else:
                w = vocab[idx[k]]
                if r[k] < 0.10:
                    w = w.capitalize()
                elif utf8 and r[k] < 0.20:
                    w = w[:1] + "ö" + w[1:]
                elif utf8 and r[k] < 0.22:
                    w = w + "€"
            s = " " if sep[k] < 0.85 else (". " if sep[k] < 0.95 else "\n")
            piece = (w + s).encode()
            if size + len(piece) > docbytes:
                break
            toks.append(piece)
            size += len(piece)
        doc = b"".join(toks)
        parts.append(doc)
        total += len(doc)
        offs.append(total)
    return b"".join(parts), np.array(offs, dtype=np.uint64)


def lexer_patterns(npatterns, vocab, seed):
    """Regex set modelled on the reference's test/doc patterns (tests/charRegexMatch :109-112, the
    rule language example of the web page): list of (lexem id, expression, resultIndex, level, posbind).
    Expressions are distinct; about 1/4 (small sets) resp. 8/10 (>1024) are \\bWORD\\b literals of
    the vocabulary (lexem id i+1 <-> vocabulary rank i+1), the rest are class / repeat / multi-word /
    alternation patterns."""
    rng = np.random.default_rng(seed)
    out = []
    seen = set()
    nlit = npatterns // 4 if npatterns <= 1024 else (npatterns * 8) // 10
    nlit = min(nlit, len(vocab))
    generic = ["[0-9]+\\b", "\\b[A-Z][a-z]+\\b", "\\b[0-9]{1,3}'[0-9]{3}\\b", "\\b[0-9]{1,2}\\b"]
    tries = 0
    while len(out) < npatterns:
        i = len(out)
        lid = i + 1
        level = 1 + int(rng.integers(0, 4))
        posbind = "content" if rng.random() < 0.85 else "predecessor"
        if i < nlit:
            expr = "\\b" + vocab[i] + "\\b"
        elif generic:
            expr = generic.pop(0)
        else:
            fam = int(rng.integers(0, 5))
            suf = "".join(_LETTERS[int(x)] for x in rng.integers(0, 14, size=int(rng.integers(2, 4))))
            if fam == 0:
                expr = "[a-z]+%s\\b" % suf
            elif fam == 1:
                expr = "\\b%s[a-z]*\\b" % suf
            elif fam == 2:
                w2 = vocab[int(rng.integers(0, min(5000, len(vocab))))]
                expr = "\\b%s\\s\\w+\\b" % w2
            elif fam == 3:
                a, b, c = (vocab[int(x)] for x in rng.integers(0, len(vocab), size=3))
                expr = "\\b(%s|%s|%s)\\b" % (a, b, c)
            else:
                expr = "\\b[A-Z]%s[a-z]*\\b" % suf[:2]
        tries += 1
        if expr in seen:
            if tries > 50 * npatterns:
                raise RuntimeError("cannot generate %d distinct expressions" % npatterns)
            continue
        seen.add(expr)
        out.append((lid, expr, 0, level, posbind))
    return out


def apply_lexer_patterns(lx, patterns, options=("DOTALL",)):
    for o in options:
        lx.defineOption(o)
    for lid, expr, residx, level, posbind in patterns:
        lx.defineLexem(lid, expr, residx, level, posbind)
    lx.compile()


def pipeline_workload(npatterns, nrules, vocab, seed):
    """Config 5 of BASELINE.json: `npatterns` regexes (plus the sentence delimiter lexem) feeding
    `nrules` two-term token rules over the lexem ids.  Returns (lexer patterns, rules)."""
    pats = lexer_patterns(npatterns, vocab, seed)
    pats.append((DELIM, "[.]", 0, 5, "content"))
    rules = random_rules(nrules, npatterns, seed + 1)
    return pats, rules


# ---------------------------------------------------------------- expression trees (BASELINE.json configs[3])
def random_tree(rng, nfeat, depth, maxdepth):
    """(range, push function) of a random expression tree: nested sequence / within / *_struct / any / sequence_imm / and
    expressions (generator after tests/randomExpressionTreeMatch/src/testRandomExpressionTreeMatch.cpp:239-338 with the depth
    limit lifted: range = argc + random + sum of the children's ranges).  push(m) issues the PatternMatcherInstanceInterface calls."""
    if depth >= maxdepth or (depth > 0 and rng.random() < 0.35):
        t = int(rng.integers(1, nfeat + 1))
        var = "t%d" % depth if rng.random() < 0.3 else None

        def push_term(m, t=t, var=var):
            m.pushTerm(t)
            if var:
                m.attachVariable(var)
        return 1, push_term
    op = ["sequence", "within", "sequence_struct", "within_struct", "any", "sequence_imm", "and"][int(rng.integers(0, 7))]
    argc = int(rng.integers(2, 4))
    children = [random_tree(rng, nfeat, depth + 1, maxdepth) for _ in range(argc)]
    rg = argc + int(rng.integers(0, 4)) + sum(c[0] for c in children)
    card = int(rng.integers(1, argc + 1)) if op in ("any", "and") and rng.random() < 0.4 else 0
    var = "e%d" % depth if depth > 0 and rng.random() < 0.3 else None

    def push(m):
        n = argc
        if op in ("sequence_struct", "within_struct"):
            m.pushTerm(DELIM)
            n += 1
        for _, c in children:
            c(m)
        m.pushExpression(op, n, rg, card)
        if var:
            m.attachVariable(var)
    return rg, push


def tree_rules(ntrees, nfeat, maxdepth, seed):
    """list of push functions, one per tree"""
    rng = np.random.default_rng(seed)
    return [random_tree(rng, nfeat, 0, maxdepth)[1] for _ in range(ntrees)]


def apply_trees(m, trees, compile=True):
    for i, push in enumerate(trees):
        push(m)
        m.definePattern("tree_%d" % i, "", True)
    if compile:
        m.compile()


def tree_documents(ndocs, n, nfeat, seed):
    """documents of n tokens over nfeat features, 5 % sentence delimiters, one token per position"""
    rng = np.random.default_rng(seed)
    lex = np.zeros((ndocs * n, 4), np.uint32)
    offs = np.arange(ndocs + 1, dtype=np.uint64) * n
    ids = rng.integers(1, nfeat + 1, size=ndocs * n)
    ids[rng.random(ndocs * n) < 0.05] = DELIM
    lex[:, 0] = ids
    lex[:, 1] = np.tile(np.arange(1, n + 1, dtype=np.uint32), ndocs)
    lex[:, 2] = np.tile(np.arange(n, dtype=np.uint32) * 2, ndocs)
    lex[:, 3] = 1
    return lex, offs
