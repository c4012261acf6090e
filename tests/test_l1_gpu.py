"""GPU parity tests of the level-1 lexer: HIP kernel (through the C-ABI) vs the CPU oracle on the
same inputs; lexem lists must be identical (id, ordpos, origpos, origsize) and in the same order.
Parity with real Hyperscan is pinned only by the 36-lexem charRegexMatch vector (first test)."""
import random
import re

import numpy as np
import pytest

import oracle
import struspattern_amd as spa
from struspattern_amd import synth
from tests import l1_cases

pytestmark = pytest.mark.gpu


def _both(build):
    lx = spa.PatternLexerInstance()
    o = oracle.L1Lexer()
    build(lx)
    build(o)
    return lx, o


def test_char_regex_match_golden_case1():
    case = l1_cases.load_char_regex_cases()[0]
    lx = spa.PatternLexerInstance()
    l1_cases.build_case(lx, case)
    got = lx.createContext().match(case["src"].encode()).tolist()
    assert got == case["result"]


@pytest.mark.parametrize("index", [1, 2])
def test_char_regex_match_golden_edit_distance_cases(index):
    """testCharRegexMatch.cpp:161-196: `abc ~1` and its UTF-8 twin, 5 lexems each"""
    case = l1_cases.load_char_regex_cases()[index]
    lx = spa.PatternLexerInstance()
    l1_cases.build_case(lx, case)
    got = lx.createContext().match(case["src"].encode()).tolist()
    assert got == case["result"]


@pytest.mark.parametrize("seed", range(6))
def test_random_approximate_literal_tables(seed):
    """Tables of plain literals with edit distances 0..3 (a table with one `~N` expression takes the approximate
    route for all of them) on random text over a small alphabet with two- and three-byte characters, batches of
    ragged documents; product vs the oracle's restatement."""
    rng = random.Random(7000 + seed)
    alphabet = ["a", "b", "c", "\u00f6", "\u00fc", "\u20ac", " "]
    chars = [c for c in alphabet if c != " "]
    npat = rng.randint(1, 5)
    defs = []
    for i in range(npat):
        n = rng.randint(2, 7)
        word = "".join(rng.choice(chars) for _ in range(n))
        dist = rng.randint(0, min(3, n - 1)) if i else rng.randint(1, min(3, n - 1))   # at least one ~N
        defs.append((i + 1, word + (" ~%d" % dist if dist else ""), rng.randint(1, 3), rng.choice(["content", "content", "predecessor", "unique"])))

    def build(x):
        for pid, expr, level, bind in defs:
            x.defineLexem(pid, expr, 0, level, bind)
        x.compile()
    lx, o = _both(build)
    docs = []
    for _ in range(40):
        n = rng.choice([0, 1, 2, 5, 30, 70, 200, 700])
        t = "".join(rng.choice(alphabet) for _ in range(n))
        for _ in range(rng.randint(0, 4)):      # plant near-matches
            w = list(rng.choice(defs)[1].split(" ~")[0])
            if w and rng.random() < 0.7:
                k = rng.randrange(len(w))
                r = rng.random()
                if r < 0.33:
                    w[k] = rng.choice(chars)
                elif r < 0.66:
                    del w[k]
                else:
                    w.insert(k, rng.choice(chars))
            at = rng.randint(0, len(t))
            t = t[:at] + "".join(w) + t[at:]
        docs.append(t.encode("utf8"))
    ctx = lx.createContext()
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    got = ctx.matchDocs(b"".join(docs), offs)
    for di, d in enumerate(docs):
        want = o.match(d).tolist()
        assert got.doc(di).tolist() == want, (defs, d)


def test_bytechar_option_on_literal_tables():
    """Option BYTECHAR (forceOneByteCharMap, patternLexer.cpp:1055-1058) sends every expression through the
    one-byte character hash and the re-match: built for tables of plain literals (hash collisions between characters
    beyond ASCII -- code points equal modulo 128 -- must be filtered by the re-match), rejected for anything else."""
    def build(x):
        x.defineOption("BYTECHAR", 0) if isinstance(x, spa.PatternLexerInstance) else x.defineOption("BYTECHAR")
        x.defineLexem(1, "a\u00f6\u00fc", 0, 1, "content")
        x.defineLexem(2, "\u00f6", 0, 2, "content")
        x.defineLexem(3, "abc", 0, 1, "content")
        x.compile()
    lx, o = _both(build)
    # U+0176 and U+01F6 hash like U+00F6 (246 mod 128), U+017C like U+00FC
    for t in ("a\u00f6\u00fc a\u0176\u017c \u01f6 \u00f6 abc abca\u00f6\u00fc", "", "\u00f6\u00f6\u00f6", "a\u0176\u00fc"):
        b = t.encode("utf8")
        assert lx.createContext().match(b).tolist() == o.match(b).tolist(), t
    lx = spa.PatternLexerInstance()
    lx.defineOption("BYTECHAR", 0)
    lx.defineLexem(1, "a+", 0, 1, "content")
    with pytest.raises(spa.PatternError):
        lx.compile()


def test_approximate_table_survives_invalid_utf8():
    def build(x):
        x.defineLexem(1, "ab\u00f6 ~1", 0, 1, "content")
        x.defineLexem(2, "\u00f6\u00f6", 0, 2, "content")
        x.compile()
    lx, o = _both(build)
    for text in (b"ab\xc3", b"\xb6ab\xc3\xb6 a\xc3\xb6\xc3", b"\xf0\x9f ab\xc3\xb6\xff\xc3\xb6\xc3\xb6", b"\x80\x80ab"):
        assert lx.createContext().match(text).tolist() == o.match(text).tolist(), text


def test_long_documents_are_scanned_in_chunks(monkeypatch):
    """A document longer than a chunk (32 KiB) is scanned as several units by several waves: the state at the start of a
    later chunk is proven from a 256-byte warm-up (from the empty state with starts injected vs. from "every position
    live" without), and a document where that proof fails -- a pattern whose state survives the warm-up -- is scanned
    again in one piece.  Lexems must not depend on where the chunks fall."""
    vocab = synth.vocabulary(3000, 3)
    pats = synth.lexer_patterns(300, vocab, 7)

    def build(x):
        synth.apply_lexer_patterns(x, pats)
    lx, o = _both(build)
    sizes = [0, 10, 200_000, 5000, 65536, 65537, 131072, 70_000]
    text, offs = synth.text_documents(len(sizes), 1000, vocab, 5, utf8=True)
    rng = random.Random(3)
    words = [w for w in vocab[:2000]]
    docs = []
    for n in sizes:
        t = []
        while sum(len(x) + 1 for x in t) < n:
            t.append(rng.choice(words) if rng.random() < 0.9 else rng.choice(["\u00e4\u00f6", "Z\u00fcrich", "12'345", "."]))
        docs.append(" ".join(t).encode("utf8")[:n])
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    ctx = lx.createContext()
    got = ctx.matchDocs(b"".join(docs), offs)
    c = ctx.batchCounters()
    assert c["scan_units"] == sum(max(1, -(-len(d) // 32768)) for d in docs) and c["rescanned_docs"] == 0
    for di, d in enumerate(docs):
        assert got.doc(di).tolist() == o.match(d).tolist(), (di, len(d))
    # every chunk boundary position: 64-byte chunks over short documents, including inside multi-byte characters and words
    monkeypatch.setenv("SPA_L1_CHUNK_BYTES", "64")
    small = [d[:n] for d in docs for n in (63, 64, 65, 127, 128, 129, 300, 1000) if len(d) >= n]
    offs = np.zeros(len(small) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in small])
    got = ctx.matchDocs(b"".join(small), offs)
    assert ctx.batchCounters()["scan_units"] > len(small)
    for di, d in enumerate(small):
        assert got.doc(di).tolist() == o.match(d).tolist(), (di, len(d))
    monkeypatch.delenv("SPA_L1_CHUNK_BYTES")

    # a state that outlives the warm-up: the chunks cannot be joined, the document is scanned again in one piece
    def build2(x):
        x.defineOption("DOTALL", 0) if isinstance(x, spa.PatternLexerInstance) else x.defineOption("DOTALL")
        x.defineLexem(1, "<[^>]*>", 0, 2, "content")
        x.defineLexem(2, "\\b\\w+\\b", 0, 1, "content")
        x.compile()
    lx2, o2 = _both(build2)
    body = (" word" * 2000).encode()                       # 10 KB without a '>' across the boundary of the first chunk
    docs2 = [b"y " * 30000 + b"<a " + body + b"> tail <b>" + b" z" * 40000, b"short <c> doc", b"xy " * 23400 + b" <d> y"]
    offs = np.zeros(len(docs2) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs2])
    ctx2 = lx2.createContext()
    # by default a table with such an expression is not scanned in chunks at all (the chunked pass would be thrown away) ...
    got = ctx2.matchDocs(b"".join(docs2), offs)
    c = ctx2.batchCounters()
    assert c["scan_units"] == len(docs2) and c["rescanned_docs"] == 0
    for di, d in enumerate(docs2):
        assert got.doc(di).tolist() == o2.match(d).tolist(), di
    # ... with chunks forced, the proof fails and the document goes through the sequential pass
    monkeypatch.setenv("SPA_L1_CHUNK_BYTES", "32768")
    got = ctx2.matchDocs(b"".join(docs2), offs)
    c = ctx2.batchCounters()
    assert c["rescanned_docs"] >= 1
    for di, d in enumerate(docs2):
        assert got.doc(di).tolist() == o2.match(d).tolist(), di
    monkeypatch.delenv("SPA_L1_CHUNK_BYTES")


def test_unicode_property_classes():
    """\\p{..}: positions classed by the decoded code point (lead bytes at every offset of the 64-byte tiles, documents
    of many tiles, malformed sequences); lexems vs the oracle."""
    def build(x):
        x.defineLexem(1, "\\b\\p{Lu}\\p{Ll}*\\b", 0, 1, "content")
        x.defineLexem(2, "\\b\\p{Ll}+\\b", 0, 1, "content")
        x.defineLexem(3, "\\p{Lu}\\p{Ll}+", 0, 2, "content")
        x.defineLexem(4, "[\\p{Nd}_]+", 0, 1, "content")
        x.defineLexem(5, "[^\\p{L}\\p{N}\\s]", 0, 1, "predecessor")
        x.defineLexem(6, "\\p{Greek}".replace("\\p{Greek}", "[\u03b1-\u03c9]+"), 0, 3, "content")
        x.compile()
    lx, o = _both(build)
    rng = random.Random(11)
    words = ["\u00c4rger", "\u00fcber", "\u00d6l", "Stra\u00dfe", "\u0391\u03b2\u03b3", "\u0416\u0443\u043a", "\u0663\u0664", "\u4f60\u597d", "\U0001d400\U0001d41a", "Abc", "x9_",
             "abc", "DEF", "!", "\u20ac", " ", " ", "\n"]
    docs = ["".join(rng.choice(words) + rng.choice(["", " "]) for _ in range(n)).encode("utf8") for n in (0, 1, 7, 60, 400, 3000)]
    docs += [b"\xc3", b"A\xc3(b", b"\xe0\x80\x80A\xed\xa0\x80b", b"\x80\xbfAb\xf4\x90\x80\x80", b"x" * 63 + "\u00c4b".encode("utf8"), b"x" * 62 + "\U0001d400b \u4f60".encode("utf8")]
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    got = lx.createContext().matchDocs(b"".join(docs), offs)
    for di, d in enumerate(docs):
        assert got.doc(di).tolist() == o.match(d).tolist(), d[:80]


def test_ucp_unicode_word_characters():
    """Option UCP on the GPU: contexts per character (lead bytes by code point, continuation bytes by their character,
    across tile boundaries), whole-word literals over Unicode word runs, start of match walking back over them."""
    def build(x):
        x.defineOption("UCP", 0) if isinstance(x, spa.PatternLexerInstance) else x.defineOption("UCP")
        x.defineLexem(1, "\\b\\w+\\b", 0, 1, "content")
        x.defineLexem(2, "\\b\\p{Lu}\\p{Ll}*\\b", 0, 2, "content")
        x.defineLexem(3, "\\d+", 0, 3, "content")
        x.defineLexem(4, "\\b\u00fcber\\b", 0, 4, "content")
        x.defineLexem(5, "\\bber\\b", 0, 4, "content")
        x.defineLexem(6, "\\B[a-z\u00df]+\\b", 0, 1, "predecessor")
        x.defineLexem(7, "\\W+", 0, 1, "predecessor")
        x.compile()
    lx, o = _both(build)
    rng = random.Random(12)
    words = ["\u00c4rger", "\u00fcber", "\u00d6l", "Stra\u00dfe", "\u0391\u03b2\u03b3", "\u0416\u0443\u043a", "\u0663\u0664", "\u4f60\u597d", "\U0001d400\U0001d41a", "Abc", "x9_",
             "ber", "DEF", "!", "\u20ac", " ", " ", "\n", "\u00a0", "\u3000"]
    docs = ["".join(rng.choice(words) + rng.choice(["", " "]) for _ in range(n)).encode("utf8") for n in (0, 1, 7, 60, 400, 3000)]
    docs += [b"\xc3", b"A\xc3(b", b"\xe0\x80\x80A\xed\xa0\x80b", b"\x80\xbfAb\xf4\x90\x80\x80", b"x" * 63 + "\u00c4b".encode("utf8"), b"x" * 62 + "\U0001d400b \u4f60".encode("utf8"),
             b"x" * 61 + "\u4f60\u597d".encode("utf8") + b"y" * 70]
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    got = lx.createContext().matchDocs(b"".join(docs), offs)
    for di, d in enumerate(docs):
        assert got.doc(di).tolist() == o.match(d).tolist(), d[:80]


def test_allowempty_option():
    """ALLOWEMPTY: empty matches become zero-length events and go through the handler like any other (restated semantics,
    pinned by no reference vector); documents of several tiles and chunks."""
    def build(x):
        x.defineOption("ALLOWEMPTY", 0) if isinstance(x, spa.PatternLexerInstance) else x.defineOption("ALLOWEMPTY")
        x.defineLexem(1, "\\b\\w+\\b", 0, 2, "content")
        x.defineLexem(2, "a*", 0, 1, "content")
        x.defineLexem(3, "x?\\b", 0, 3, "predecessor")
        x.defineLexem(4, "[0-9]+", 0, 2, "content")
        x.compile()
    lx, o = _both(build)
    rng = random.Random(21)
    docs = [b"", b"a", b"baab aa", b" x1 22y "] + ["".join(rng.choice("ab x1 .") for _ in range(n)).encode() for n in (70, 300, 2000)]
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    got = lx.createContext().matchDocs(b"".join(docs), offs)
    for di, d in enumerate(docs):
        assert got.doc(di).tolist() == o.match(d).tolist(), d[:60]


def test_wide_alternations_cut_into_several_words():
    """Expressions beyond 64 byte positions (cut at an alternation into several automaton words): several words of
    one expression report at the same end offset with different starts (suffix-related alternatives sit in
    different words), on enough text that groups of such reports meet the 64-report batch boundaries."""
    words = ["w%02d%s" % (i, "xyz"[i % 3] * (1 + i % 4)) for i in range(30)]
    alts = ["abcdefgh"] + words[:15] + ["cdefgh"] + words[15:] + ["fgh", "h"]

    def build(x):
        x.defineLexem(1, "(%s)" % "|".join(alts), 0, 2, "content")
        x.defineLexem(2, "[a-z0-9]+", 0, 1, "content")
        x.defineLexem(3, "q(%s)\\b" % "|".join(alts), 0, 3, "content")
        x.compile()
    lx, o = _both(build)
    rng = random.Random(5)
    pieces = ["abcdefgh", "xcdefgh", "fgh", "h", "qabcdefghfgh", " ", " ", "zz"] + words
    docs = [" ".join(rng.choice(pieces) for _ in range(n)).encode() for n in (0, 1, 3, 40, 300, 1500)]
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    got = lx.createContext().matchDocs(b"".join(docs), offs)
    for di, d in enumerate(docs):
        assert got.doc(di).tolist() == o.match(d).tolist(), d[:80]


def test_supersede_levels_symbols():
    def build(x):
        x.defineLexem(1, "\\b\\w+\\b", 0, 1, "content")
        x.defineLexem(2, "[.]", 0, 2, "content")
        x.defineLexem(3, "\\b[a-z]+[.][a-z]+\\b", 0, 3, "content")
        x.defineLexem(4, "[0-9]+", 0, 1, "successor")
        x.defineLexem(5, "\\b[a-z]{3}\\b", 0, 1, "unique")
        x.defineSymbol(70, 1, "now")
        x.defineSymbol(71, 1, "go")
        x.compile()
    lx, o = _both(build)
    text = b"go to example.com now. call 555 1234 now or see foo.bar.baz for the cat"
    assert lx.createContext().match(text).tolist() == o.match(text).tolist()


@pytest.mark.parametrize("seed", range(4))
def test_random_regex_sets(seed):
    rng = random.Random(5000 + seed)
    ctxs = []
    for _ in range(12):
        pats = []
        while len(pats) < rng.randint(1, 8):
            p = l1_cases.random_regex(rng)
            try:
                re.compile(p)
                one = spa.PatternLexerInstance()
                one.defineOption("DOTALL")
                one.defineLexem(1, p, 0, 1, "content")
                one.compile()
            except (re.error, spa.PatternError):
                continue
            pats.append((p, rng.randint(1, 3), rng.choice(["content", "content", "predecessor", "successor", "unique"])))

        def build(x):
            x.defineOption("DOTALL")
            for i, (p, level, pb) in enumerate(pats):
                x.defineLexem(1 + i % 5, p, 0, level, pb)
            x.compile()
        lx, o = _both(build)
        docs = [l1_cases.random_text(rng, rng.randint(0, 200)).encode() for _ in range(20)]
        offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
        text = b"".join(docs)
        gpu = lx.createContext().matchDocs(text, offs)
        ref, roffs = o.matchDocs(text, offs)
        assert np.array_equal(gpu.doc_offsets, roffs), pats
        assert np.array_equal(gpu.lexems, ref), pats


# shapes=True: the default -- whole-word literals and word shapes (l1_tables.h) by the words kernel, the rest by the scan kernel;
# shapes=False (SPA_L1_SHAPES=0): every expression that is not a literal in the scanned passes -- the (11000, ..) case then compiles
# to 3 automaton passes (a pass count that is not a power of two), the (4800, ..) case fills one pass only when packed by size and
# the kernel sorts the reports of an end offset
@pytest.mark.parametrize("shapes", [True, False])
@pytest.mark.parametrize("npat,ndocs,docbytes,utf8,seed", [(64, 16, 2000, False, 1), (256, 24, 4096, False, 2), (256, 8, 3000, True, 3), (700, 6, 2000, False, 4),
                                                            (11000, 4, 2000, False, 5),
                                                            (4800, 12, 3000, False, 6)])
def test_synthetic_lexer_workload(npat, ndocs, docbytes, utf8, seed, shapes, monkeypatch):
    from tests.l1_table_sim import Tables
    if not shapes:
        monkeypatch.setenv("SPA_L1_SHAPES", "0")
    vocab = synth.vocabulary(12000 if npat > 6000 else 6000 if npat > 3000 else 3000, 77)
    pats = synth.lexer_patterns(npat, vocab, seed)
    text, offs = synth.text_documents(ndocs, docbytes, vocab, 100 + seed, utf8=utf8)
    lx, o = _both(lambda x: synth.apply_lexer_patterns(x, pats))
    t = Tables(lx.dumpTables())
    if not shapes:
        assert t.nof_shapes == 0 and t.scan_passes == t.npasses
        if npat == 4800:
            assert int(lx.dumpTables()[6]) == 0 and t.npasses == 1
        if npat == 11000:
            assert t.npasses == 3
    elif npat >= 700:
        assert t.nof_shapes > npat // 20 and t.scan_passes == 1 and t.npasses > t.scan_passes
    gpu = lx.createContext().matchDocs(text, offs)
    ref, roffs = o.matchDocs(text, offs, nthreads=8)
    assert len(ref) > 100
    assert np.array_equal(gpu.status, np.zeros(ndocs, np.int32))
    assert np.array_equal(gpu.doc_offsets, roffs)
    assert np.array_equal(gpu.lexems, ref)


def test_edge_cases_empty_documents_and_errors():
    def build(x):
        x.defineLexem(1, "[a-z]+\\b", 0, 1, "content")
        x.compile()
    lx, o = _both(build)
    docs = [b"", b"a", b"", b"hello world", b" ", b""]
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    gpu = lx.createContext().matchDocs(b"".join(docs), offs)
    ref, roffs = o.matchDocs(b"".join(docs), offs)
    assert np.array_equal(gpu.doc_offsets, roffs) and np.array_equal(gpu.lexems, ref)
    # a lexem of 65535 bytes or more is an error (patternLexer.cpp:727-730)
    big = b"a" * 70000
    ctx = lx.createContext()
    b = ctx.matchDocs(big + b" ok", [0, len(big) + 3], check=False)
    assert b.status[0] == 7
    with pytest.raises(spa.PatternError):
        ctx.match(big)
    # context before compile is an error (patternLexer.cpp:1124-1127)
    lx2 = spa.PatternLexerInstance()
    lx2.defineLexem(1, "a", 0, 1, "content")
    with pytest.raises(spa.PatternError):
        lx2.createContext()


# The handler of the post-processing kernel takes a cluster of reports per lane (postDocumentClusters); this set makes the cases it
# hands back to the one-report-after-the-other path: more survivors on one word than a lane's event array holds, symbol lookups,
# chains of two-word matches that tie a whole line into one cluster, sub expression selection that moves the end of a match.
@pytest.mark.parametrize("mode", ["clusters", "sequential", "clusters-chunked"])
def test_handler_clusters_against_the_oracle(mode, monkeypatch):
    if mode == "sequential":
        monkeypatch.setenv("SPA_L1_POST_SEQ", "1")
    if mode == "clusters-chunked":
        monkeypatch.setenv("SPA_L1_CHUNK_BYTES", "1024")
    rng = random.Random(99)
    words = ["banana", "bandana", "cabana", "aa", "ab", "nab", "bank", "an", "na", "ana", "band", "a", "b", "kab", "abba", "nanana"]
    posbinds = ["content", "content", "content", "predecessor", "successor", "unique"]

    def build(x):
        x.defineOption("DOTALL")
        lid = 1
        # every prefix and suffix shape of "banana" at one level: seven and more events survive on that word
        for k in range(1, 6):
            x.defineLexem(lid, "\\b%s[a-z]*\\b" % "banana"[:k], 0, 2, "content"); lid += 1
            x.defineLexem(lid, "[a-z]+%s\\b" % "banana"[-k:], 0, 2, "content"); lid += 1
        r = random.Random(7)
        for w in words:
            x.defineLexem(lid, "\\b%s\\b" % w, 0, r.randint(1, 4), r.choice(posbinds)); lid += 1
        for w in ("aa", "ab", "an", "banana"):
            x.defineLexem(lid, "\\b%s\\s\\w+\\b" % w, 0, r.randint(1, 4), r.choice(posbinds)); lid += 1
        for suf in ("b", "na", "nd", "k"):
            x.defineLexem(lid, "[a-z]+%s\\b" % suf, 0, r.randint(1, 4), r.choice(posbinds)); lid += 1
        for pre in ("ca", "n", "ab"):
            x.defineLexem(lid, "\\b%s[a-z]*\\b" % pre, 0, r.randint(1, 4), r.choice(posbinds)); lid += 1
        x.defineLexem(lid, "[a-z]+[.][a-z]+", 0, 3, "content"); lid += 1          # (automaton queue)
        x.defineLexem(lid, "[0-9]+", 0, 1, "successor"); lid += 1
        x.defineLexem(lid, "\\b(x)([a-z]+)(y)\\b", 2, 2, "content"); lid += 1      # sub expression selection
        x.defineLexem(lid, "\\b\\w+\\b", 0, 1, "content")
        x.defineSymbol(900, lid, "bank")
        x.defineSymbol(901, lid, "aa")
        x.compile()
    lx, o = _both(build)
    docs = []
    for d in range(40):
        toks = []
        for _ in range(rng.randint(0, 260)):
            u = rng.random()
            if u < 0.75:
                toks.append(rng.choice(words))
            elif u < 0.80:
                toks.append(str(rng.randint(0, 999)))
            elif u < 0.85:
                toks.append("x" + rng.choice(words) + "y")
            elif u < 0.90:
                toks.append(rng.choice(words) + "." + rng.choice(words))
            else:
                toks.append("aa " * rng.randint(2, 90))               # (one cluster longer than a window)
            toks.append(rng.choice([" ", " ", " ", ". ", "\n", ", "]))
        docs.append("".join(toks).encode())
    docs.append(("aa " * 6000).encode())
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    text = b"".join(docs)
    ctx = lx.createContext()
    gpu = ctx.matchDocs(text, offs)
    assert ctx.batchCounters()["word_reports"] > 1000
    ref, roffs = o.matchDocs(text, offs, nthreads=8)
    assert len(ref) > 5000
    assert np.array_equal(gpu.status, np.zeros(len(docs), np.int32))
    assert np.array_equal(gpu.doc_offsets, roffs)
    assert np.array_equal(gpu.lexems, ref)


# The words kernel collects run ends over several tiles and probes them 64 at a time (wordsFlush); this set drives its corners: tiles
# where every second byte ends a run (32 ends per tile, several candidates each: more than 64 candidates per batch, several rounds of
# walks), ends so sparse that the batch is flushed because the 512-byte ring is about to lose the bytes before them, words longer
# than the ring (the confirming walk leaves it), table entries that hold several patterns (list merge instead of the sorting network).
@pytest.mark.parametrize("chunk", [None, "2048"])
def test_words_kernel_batches_against_the_oracle(chunk, monkeypatch):
    if chunk:
        monkeypatch.setenv("SPA_L1_CHUNK_BYTES", chunk)
    rng = random.Random(4242)

    def build(x):
        x.defineOption("DOTALL")
        lid = 1
        for pre in ("a", "ab", "b", "ba", "abc"):
            x.defineLexem(lid, "\\b%s[a-z]*\\b" % pre, 0, 1 + lid % 3, "content"); lid += 1
        for suf in ("a", "b", "ab", "ba", "cab"):
            x.defineLexem(lid, "[a-z]+%s\\b" % suf, 0, 1 + lid % 3, "content"); lid += 1
        # the same shape key twice (one table entry, a list of two patterns), the same word in a literal and in two alternations
        x.defineLexem(lid, "\\bab[a-z]*\\b", 0, 3, "predecessor"); lid += 1
        x.defineLexem(lid, "\\b[a-z]+ab\\b", 0, 2, "content"); lid += 1
        x.defineLexem(lid, "\\bab\\b", 0, 2, "content"); lid += 1
        x.defineLexem(lid, "\\b(ab|ba|zz)\\b", 0, 1, "content"); lid += 1
        x.defineLexem(lid, "\\b(ab|a|b)\\b", 0, 4, "successor"); lid += 1
        for w in ("a", "ab", "b"):
            x.defineLexem(lid, "\\b%s\\s\\w+\\b" % w, 0, 1 + lid % 4, "content"); lid += 1
        x.defineLexem(lid, "[0-9]+[.][0-9]+", 0, 2, "content"); lid += 1      # (keeps an automaton pass to scan)
        x.compile()
    lx, o = _both(build)
    from tests.l1_table_sim import Tables
    assert Tables(lx.dumpTables()).nof_shapes >= 8
    docs = []
    docs.append(("a b " * 3000).encode())                                            # 32 run ends per tile
    docs.append(("ab ba " * 2500).encode())
    docs.append(b"".join(b"ab" + b" " * rng.randint(300, 900) for _ in range(60)))  # sparse ends
    docs.append(("ab" + "c" * 700 + "ab " + "b" * 600 + "a" + " " * 40).encode() * 6)  # words longer than the ring
    for _ in range(12):
        toks = []
        for _ in range(rng.randint(50, 700)):
            u = rng.random()
            if u < 0.6:
                toks.append(rng.choice(["a", "b", "ab", "ba", "abc", "cab", "abab", "zz", "bab", "abcab"]))
            elif u < 0.7:
                toks.append("%d.%d" % (rng.randint(0, 99), rng.randint(0, 99)))
            else:
                toks.append("".join(rng.choice("abc") for _ in range(rng.randint(1, 30))))
            toks.append(rng.choice([" ", " ", "  ", ". ", "\n", " " * rng.randint(1, 200)]))
        docs.append("".join(toks).encode())
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    text = b"".join(docs)
    ctx = lx.createContext()
    gpu = ctx.matchDocs(text, offs)
    assert ctx.batchCounters()["word_reports"] > 10000
    ref, roffs = o.matchDocs(text, offs, nthreads=8)
    assert np.array_equal(gpu.status, np.zeros(len(docs), np.int32))
    assert np.array_equal(gpu.doc_offsets, roffs)
    assert np.array_equal(gpu.lexems, ref)
