"""CPU-side tests of the product's regex compiler (no GPU): the compiled bit-parallel tables,
executed by the pure-Python table interpreter in tests/l1_table_sim.py, must produce the same raw
report stream (pattern, leftmost start, end) as the CPU oracle and as Python's `re`."""
import random
import re

import pytest

import oracle
import struspattern_amd as spa
from tests import l1_cases
from tests.l1_table_sim import Tables


def _product_reports(patterns, text, options=("DOTALL",)):
    lx = spa.PatternLexerInstance()
    for o in options:
        lx.defineOption(o)
    for i, p in enumerate(patterns):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    return Tables(lx.dumpTables()).raw_reports(text)


def _oracle_reports(patterns, text, options=("DOTALL",)):
    lx = oracle.L1Lexer()
    for o in options:
        lx.defineOption(o)
    for i, p in enumerate(patterns):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    raw, _ = lx.matchDocs(text, [0, len(text)], raw=True)
    return [(int(r[0]), int(r[1]), int(r[2])) for r in raw]


def test_golden_patterns_raw_reports():
    case = l1_cases.load_char_regex_cases()[0]
    pats = [p[1] for p in case["patterns"]]
    text = case["src"].encode()
    assert _product_reports(pats, text) == _oracle_reports(pats, text)


@pytest.mark.parametrize("seed", range(8))
def test_random_regex_tables_vs_oracle_and_python_re(seed):
    rng = random.Random(1000 + seed)
    for _ in range(25):
        pats = []
        while len(pats) < rng.randint(1, 6):
            p = l1_cases.random_regex(rng)
            try:
                re.compile(p)
            except re.error:
                continue
            try:        # documented limit of this version: 64 byte positions per expression
                one = spa.PatternLexerInstance()
                one.defineOption("DOTALL")
                one.defineLexem(1, p, 0, 1, "content")
                one.compile()
            except spa.PatternError as e:
                assert "too complex" in str(e) or "matches empty buffer" in str(e), str(e)
                continue
            pats.append(p)
        text = l1_cases.random_text(rng, rng.randint(0, 30)).encode()
        got = _product_reports(pats, text)
        assert got == _oracle_reports(pats, text), (pats, text)
        for i, p in enumerate(pats):
            exp = l1_cases.py_leftmost_reports(p, text.decode(), re.DOTALL | re.ASCII)
            assert [(f, t) for (k, f, t) in got if k == i + 1] == exp, (p, text)


def test_options_and_utf8():
    cases = [
        (["abc"], "xABCx", ("CASELESS",)), (["a.c"], "a\nc", ()), (["a.c"], "a\nc", ("DOTALL",)),
        (["^b", "a$"], "a\nb", ("MULTILINE",)), (["^b", "a$", "b$"], "a\nb", ()),
        (["a.c", "[^x]+", "ö+", "[ä-ü]", "\\W+", "\\w+\\b"], "aöc öö x", ()),
        (["[0-9]{1,3}'[0-9]{3}\\b", "\\b[A-Z][a-z]+\\b", "(Mr|Mrs|Dr)\\.\\s[A-Z][a-z]+"], "Dr. Who paid 12'345 to Mrs. Smith", ()),
    ]
    for pats, text, opts in cases:
        t = text.encode()
        assert _product_reports(pats, t, opts) == _oracle_reports(pats, t, opts), (pats, text, opts)


def test_unicode_property_classes():
    """\\p{..} (general categories from Python's unicodedata, tools/gen_unicode_tables.py): a large code point set is
    one automaton position entered at the lead byte and classed by the decoded code point; malformed sequences
    (truncated, overlong, surrogates, stray continuation bytes) fall back to byte classes and match no property."""
    pats = ["\\b\\p{Lu}\\p{Ll}*\\b", "\\b\\p{Ll}+\\b", "[\\p{Nd}x]+", "\\P{L}+", "[^\\p{L}\\s]", "\\p{Lu}\\p{Ll}+", "\\pL\\p{^L}", "[\u00e4\u00f6]\\p{Greek}?".replace("\\p{Greek}?", "")]
    texts = ["\u00c4rger \u00fcber \u00d6l und Stra\u00dfe 123x \u0391\u0392\u03b3\u03b4 \u0416\u0443\u043a \u0663\u0664 \u4f60\u597d \U0001d400\U0001d41a!",
             "abc DEF Ghi", "\u00e9\u00c9x", "", "A", "\u00df"]
    for t in texts:
        b = t.encode("utf8")
        assert _product_reports(pats, b) == _oracle_reports(pats, b), t
    for b in (b"\xc3", b"A\xc3(b", b"\xe0\x80\x80A\xed\xa0\x80b", b"\x80\xbfAb\xf4\x90\x80\x80", b"\xc3\x84\xc3", b"\xf0\x9d\x90"):
        assert _product_reports(pats, b) == _oracle_reports(pats, b), b
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "\\p{Nope}", 0, 1, "content")
    with pytest.raises(spa.PatternError):
        lx.compile()


def test_ucp_unicode_word_characters():
    """Option UCP: \\w \\d \\s by Unicode properties, \\b / \\B between characters by whether they are word
    characters (L | N | _): every byte of a multi-byte character carries its character's context."""
    pats = ["\\b\\w+\\b", "\\b\\p{Lu}\\p{Ll}*\\b", "\\d+", "\\s+", "\\B[a-z\u00df]", "[\u00e4\u00f6\u00fc]\\b", "x\\b.", "\\b\u00e9", "\\bber\\b", "\\b\u00fcber\\b", "\\W+"]
    opts = ("DOTALL", "UCP")
    texts = ["\u00c4rger \u00fcber \u00d6l und Stra\u00dfe 123x \u0663\u0664 \u0391\u0392\u03b3\u03b4 \u0416\u0443\u043a x y ber",
             "x\u00e9 \u00e9x x.\u00e9 \u20ac\u00e9", "\u65e5\u672c\u8a9e text\u3000end", "", "\U0001d400\U0001d41ab \U0001f600x", "a\u0085b\u00a0c"]
    for t in texts:
        b = t.encode("utf8")
        assert _product_reports(pats, b, opts) == _oracle_reports(pats, b, opts), t
    for b in (b"\xc3", b"A\xc3(b", b"\xe0\x80\x80A\xed\xa0\x80b", b"\x80\xbfAb\xf4\x90\x80\x80", b"\xc3\x84\xc3 x\xc3\xa9\xa9", b"ab\xf0\x9d\x90"):
        assert _product_reports(pats, b, opts) == _oracle_reports(pats, b, opts), b
    # without the option the same expressions see ASCII word characters only
    assert _product_reports(["\\b\\w+\\b"], "\u00fcber".encode("utf8")) == [(1, 2, 5)]
    assert _product_reports(["\\b\\w+\\b"], "\u00fcber".encode("utf8"), opts) == [(1, 0, 5)]


def test_allowempty_reports_empty_matches():
    """Option ALLOWEMPTY (HS_FLAG_ALLOWEMPTY): an expression that can match the empty string also reports (offset, offset)
    wherever its empty path holds and nothing longer of it ends -- restated from Hyperscan's documentation, pinned by no
    vector of the reference."""
    pats = ["a*", "\\b", "x?\\b", "(?:ab)*c?", "\\B", "^", "$", "[0-9]+"]
    opts = ("DOTALL", "ALLOWEMPTY")
    for t in (b"baab aa", b"", b"a", b" x1 22y ", b"abab c"):
        assert _product_reports(pats, t, opts) == _oracle_reports(pats, t, opts), t
    got = _product_reports(["a*"], b"baab", opts)
    assert got == [(1, 0, 0), (1, 1, 1), (1, 1, 2), (1, 1, 3), (1, 4, 4)]


def test_caseless_folds_beyond_ascii():
    """CASELESS in UTF-8 mode uses Unicode case classes (tools/gen_unicode_tables.py): literals, classes, ranges."""
    pats = ["stra\u00dfe", "[a-z]+k", "\u00c4\u00d6[\u00fc]+", "\u03a3\u03af\u03c3\u03c5\u03c6\u03bf\u03c2", "[\u0430-\u044f]+", "x\u017f"]
    texts = ["STRASSE Stra\u00dfe STRA\u00dfE stra\u1e9ee", "abc\u212a xyzK", "\u00e4\u00f6\u00dc\u00fc\u00dc \u00c4\u00d6\u00fc", "\u03c3\u03af\u03c3\u03c5\u03c6\u03bf\u03c3 \u03a3\u038a\u03a3\u03a5\u03a6\u039f\u03a3",
             "\u041f\u0440\u0438\u0432\u0435\u0442 \u043c\u0438\u0440", "XS xs x\u017f Xs"]
    for t in texts:
        b = t.encode("utf8")
        assert _product_reports(pats, b, ("DOTALL", "CASELESS")) == _oracle_reports(pats, b, ("DOTALL", "CASELESS")), t
    assert len(_product_reports(pats, texts[3].encode("utf8"), ("DOTALL", "CASELESS"))) > 0
    assert len(_product_reports(pats, texts[4].encode("utf8"), ("DOTALL", "CASELESS"))) > 0


def test_wide_alternations_are_cut_into_several_words():
    """An expression of more than 64 byte positions is cut at an alternation into entries of one 64-bit word each
    (same definition index); their reports are merged into one per end offset with the leftmost start."""
    tlds = "aero|asia|biz|cat|com|coop|edu|gov|info|int|jobs|mil|mobi|museum|name|net|org|pro|tel|travel|ac|ad|ae|af|ag|ai|al|am|an|ao|aq|ar|as|at|au|aw|ax|az|ba|bb|bd|be|ch|de|uk|us"
    pats = ["([^\\s/?\\.#-][^\\s/?\\.#-]+\\.)(%s)" % tlds, "\\b\\w+\\b", "x(%s)y|z(%s)" % (tlds, tlds)]
    lx = spa.PatternLexerInstance()
    for i, p in enumerate(pats):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    T = Tables(lx.dumpTables())
    assert len(T.patterns) > len(pats) and sorted(set(p["defIndex"] for p in T.patterns)) == [0, 1, 2]
    for text in (b"see www.etc.ch or mail.museum.travel, not a.b; xaeroy zcom xxbey www.example.com/x?y",
                 b"a.ch", b"ab.ch.de.uk", b"", b"xmuseumy.aero zaero.info.infox"):
        assert _product_reports(pats, text) == _oracle_reports(pats, text), text
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "a" * 70, 0, 1, "content")       # nothing to cut at
    with pytest.raises(spa.PatternError):
        lx.compile()


def test_compile_errors_are_reported():
    for bad in ["(abc", "abc)", "[abc", "a{3,1}", "*a", "\\p{Lx}", "(?=a)b", "a\\"]:
        lx = spa.PatternLexerInstance()
        lx.defineLexem(1, bad, 0, 1, "content")
        with pytest.raises(spa.PatternError):
            lx.compile()
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "a[bc]d ~1", 0, 1, "content")  # edit distance on anything but a plain literal: rejected loudly
    with pytest.raises(spa.PatternError):
        lx.compile()
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "abc ~1", 0, 1, "content")     # ... and a regex beside an edit distance literal as well
    lx.defineLexem(2, "x+", 0, 1, "content")
    with pytest.raises(spa.PatternError):
        lx.compile()
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "abc ~1", 0, 1, "content")     # approximate literal table (testCharRegexMatch.cpp:161-196)
    lx.compile()
    lx = spa.PatternLexerInstance()
    lx.defineLexem(1, "abc", 0, 1, "content")
    lx.compile()
    with pytest.raises(spa.PatternError):
        lx.defineLexem(2, "x", 0, 1, "content")       # define after compile (patternLexer.cpp:999-1002)
    lx = spa.PatternLexerInstance()
    with pytest.raises(spa.PatternError):
        lx.defineOption("NOPE")
    with pytest.raises(spa.PatternError):
        lx.defineLexem(1 << 30, "a", 0, 1, "content")  # id out of range (:80-83)
    with pytest.raises(spa.PatternError):
        lx.defineLexem(1, "a", 0, 256, "content")      # level out of range (:84-87)


def test_symbols_and_names():
    lx = spa.PatternLexerInstance()
    lx.defineLexemName(1, "WORD")
    assert lx.getLexemName(1) == "WORD" and lx.getLexemName(2) is None
    lx.defineLexem(1, "[a-z]+\\b", 0, 1, "content")
    lx.defineSymbol(7, 1, "cat")
    assert lx.getSymbol(1, "cat") == 7 and lx.getSymbol(1, "dog") == 0
    with pytest.raises(spa.PatternError):
        lx.defineSymbol(8, 1, "cat")                   # symbol defined twice (:286-290)


def test_size_ordered_packing_reports_the_same(monkeypatch):
    """4800 synthetic patterns need two passes in definition order and one when packed by size; the raw
    reports (sorted by end offset, pattern index) must not depend on the packing.  (Every expression in the
    scanned passes: SPA_L1_SHAPES=0.)"""
    from struspattern_amd import synth
    monkeypatch.setenv("SPA_L1_SHAPES", "0")
    vocab = synth.vocabulary(6000, 77)
    pats = synth.lexer_patterns(4800, vocab, 6)
    text, offs = synth.text_documents(1, 1500, vocab, 106, utf8=False)
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    dump = lx.dumpTables()
    assert int(dump[0]) == 1 and int(dump[6]) == 0
    o = oracle.L1Lexer()
    synth.apply_lexer_patterns(o, pats)
    raw, _ = o.matchDocs(text, [0, len(text)], raw=True)
    assert Tables(dump).raw_reports(text) == [(int(r[0]), int(r[1]), int(r[2])) for r in raw]


def test_word_shapes_keep_their_automata_behind_the_scanned_passes():
    """Expressions pinned by a few literal bytes of a word run (l1_tables.h: PREFIX / SUFFIX / PREVWORD) are found by the
    words kernel; they keep their automaton positions -- the backward walk that confirms a candidate needs them -- in passes
    of their own: all passes together still report what the oracle reports, and the scanned passes lose the shapes."""
    from struspattern_amd import synth
    vocab = synth.vocabulary(6000, 77)
    pats = synth.lexer_patterns(4800, vocab, 6)
    text, offs = synth.text_documents(1, 1500, vocab, 106, utf8=False)
    lx = spa.PatternLexerInstance()
    synth.apply_lexer_patterns(lx, pats)
    dump = lx.dumpTables()
    t = Tables(dump)
    o = oracle.L1Lexer()
    synth.apply_lexer_patterns(o, pats)
    raw, _ = o.matchDocs(text, [0, len(text)], raw=True)
    assert t.raw_reports(text) == [(int(r[0]), int(r[1]), int(r[2])) for r in raw]
    nshape = sum(1 for _, e, _, _, _ in pats if e.startswith("[a-z]+") or ("\\s\\w+" in e) or e.endswith("[a-z]*\\b"))
    assert t.scan_passes < t.npasses and t.nof_shapes == nshape and nshape > 700


def test_classes_covering_all_non_ascii_characters():
    """'.' and negated ASCII classes take the compact 6-position form for "any character beyond ASCII";
    on valid UTF-8 of every sequence length (incl. the first and last code point of each length) the
    reports must equal the oracle's, which splits the ranges exactly."""
    text = ("a\u0080b\u07ffc\u0800d\uffffe\U00010000f\U0010ffffg x\u00e9y \u20ac\u20ac z").encode("utf-8")
    for pats in (["a.b", "[^a-z ]+", "\\b[^\\s]+\\b", ".", "[^\\x00-\\x7f]{2}", "x[^q]y"], ["[^.]{3}", "(?:.|q){2}z"]):
        assert _product_reports(pats, text) == _oracle_reports(pats, text), pats
        assert _product_reports(pats, text, options=()) == _oracle_reports(pats, text, options=()), pats


SHARED_HEAD_PATTERNS = [
    "[a-z]+ing\\b", "[a-z]+ed\\b", "[a-z]+s\\b", "[a-z]+ings\\b",        # one looping first position for four patterns
    "\\bun[a-z]+\\b", "\\bup[a-z]*\\b", "\\bunder\\s\\w+\\b", "\\bu\\b",    # 'u' after a word boundary
    "\\b[A-Z]an[a-z]*\\b", "\\b[A-Z][a-z]+\\b", "\\b[A-Z]\\.",               # [A-Z] after a word boundary
    "a+b", "a+c", "ab", "ac", "a+",                                        # looping and plain 'a' are different heads; "a+" accepts in its head
    "(ab)+c", "(ab)+d",                                                    # an edge leads back into the first position: not shared
    "[0-9]+th\\b", "[0-9]+st\\b", "[0-9]+\\b",
    "x[0-9]{1,3}y", "x[0-9]{2}z",                                          # members with exception edges of their own
]
SHARED_HEAD_TEXT = (b"Running under water the singer sings songs and tested beds. Uncle Dan and Ann upped the "
                    b"unders, up u un. A. B.C aaab aab ab ac abababc ababd a 5th 21st 7 x1y x12y x123y x12z x1234y "
                    b"undertaking  understood upkeep Andes Dans wings")


def test_patterns_sharing_their_first_position(monkeypatch):
    """SPA_L1_SHARE=force packs patterns with an identical first position onto one shared bit whenever they
    qualify; the raw reports (end offset, pattern, leftmost start) must be those of the unshared tables and
    of the oracle."""
    expected = _oracle_reports(SHARED_HEAD_PATTERNS, SHARED_HEAD_TEXT)
    assert len(expected) > 60
    monkeypatch.setenv("SPA_L1_SHARE", "off")
    lx = spa.PatternLexerInstance()
    for i, p in enumerate(SHARED_HEAD_PATTERNS):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    plain = lx.dumpTables()
    assert Tables(plain).raw_reports(SHARED_HEAD_TEXT) == expected
    monkeypatch.setenv("SPA_L1_SHARE", "force")
    lx = spa.PatternLexerInstance()
    for i, p in enumerate(SHARED_HEAD_PATTERNS):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    shared = lx.dumpTables()
    # 4+4+3+2+2+3(th/st: "[0-9]+\\b" accepts in its head)... the exact count is the layout's business: fewer positions, same reports
    assert int(shared[4]) <= int(plain[4]) - 10 and int(shared[6]) == 0
    assert Tables(shared).raw_reports(SHARED_HEAD_TEXT) == expected


def test_sharing_the_first_position_saves_a_pass(monkeypatch):
    """1150 suffix / prefix patterns need two passes bit for bit and one with shared first positions
    (SPA_L1_SHARE=on takes that layout when it saves a pass); the reports stay those of the oracle."""
    import itertools
    letters = "abcdefghijklmnopqrstuvwxyz"
    sufs = ["".join(t) for t in itertools.product(letters[:12], letters[:10], letters[:5])]
    pats = ["[a-z]+%s\\b" % s for s in sufs[:520]] + ["\\b%s[a-z]*\\b" % s for s in sufs[40:560]] + ["\\b[A-Z]%s[a-z]*\\b" % s[:2] for s in sufs[::5][:110]]
    pats = list(dict.fromkeys(pats))
    monkeypatch.setenv("SPA_L1_SHARE", "on")
    lx = spa.PatternLexerInstance()
    for i, p in enumerate(pats):
        lx.defineLexem(i + 1, p, 0, 1, "content")
    lx.compile()
    dump = lx.dumpTables()
    assert int(dump[0]) == 1 and int(dump[6]) == 0 and int(dump[4]) < 4096
    monkeypatch.setenv("SPA_L1_SHARE", "off")
    lx2 = spa.PatternLexerInstance()
    for i, p in enumerate(pats):
        lx2.defineLexem(i + 1, p, 0, 1, "content")
    lx2.compile()
    assert int(lx2.dumpTables()[0]) == 2
    text = (" ".join(sufs[k] + "x" + sufs[(7 * k) % 600] + " " + sufs[(3 * k) % 600].capitalize() for k in range(0, 600, 13))).encode()
    assert Tables(dump).raw_reports(text) == _oracle_reports(pats, text, options=())
