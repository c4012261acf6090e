"""Shared L1 test inputs: the reference's charRegexMatch known answers as data + builders driving
any object with the PatternLexerInstance method names, and random regex/text generators for the
cross-checks."""
import json
import os
import random

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_char_regex_cases():
    with open(os.path.join(GOLDEN, "char_regex_match.json")) as f:
        return json.load(f)["cases"]


def build_case(lx, case):
    """compile() of testCharRegexMatch.cpp:66-85 after defineOption("DOTALL") (:236)."""
    lx.defineOption("DOTALL", 0)
    for pid, expr, residx, level, haspos in case["patterns"]:
        lx.defineLexem(pid, expr, residx, level, "content" if haspos else "predecessor")
    for symid, patid, name in case["symbols"]:
        lx.defineSymbol(symid, patid, name)
    lx.compile()


# ---- random regexes over a tiny alphabet for differential tests against Python's `re`
ATOMS = ["a", "b", "c", "ab", "[ab]", "[^a]", ".", "\\w", "\\s", "\\d", "[a-c]", "x", " ", "1", "\\W"]


def random_regex(rng, depth=0, quant_ok=True):
    """Random regex over a tiny alphabet.  Quantifiers are never nested (a quantified group contains
    no further quantifier): nested unbounded repeats make the backtracking cross-check exponential."""
    r = rng.random()
    quantify = quant_ok and rng.random() < 0.5
    inner_ok = quant_ok and not quantify
    if depth > 2 or r < 0.35:
        a = rng.choice(ATOMS)
    elif r < 0.55:
        a = "(" + random_regex(rng, depth + 1, inner_ok) + "|" + random_regex(rng, depth + 1, inner_ok) + ")"
    elif r < 0.7:
        a = "(?:" + random_regex(rng, depth + 1, inner_ok) + ")"
    else:
        a = random_regex(rng, depth + 1, inner_ok) + random_regex(rng, depth + 1, inner_ok)
    if quantify:
        q = rng.random()
        if q < 0.25:
            a = "(?:" + a + ")*"
        elif q < 0.55:
            a = "(?:" + a + ")+"
        elif q < 0.75:
            a = "(?:" + a + ")?"
        elif q < 0.9:
            a = "(?:" + a + "){1,3}"
        else:
            a = "(?:" + a + "){2}"
    if depth == 0:
        if rng.random() < 0.3:
            a = "\\b" + a
        if rng.random() < 0.4:
            a = a + "\\b"
        if rng.random() < 0.05:
            a = "^" + a
        if rng.random() < 0.05:
            a = a + "$"
    elif rng.random() < 0.05:
        a = a + "\\b"
    elif rng.random() < 0.03:
        a = "\\B" + a
    return a


def random_text(rng, n):
    return "".join(rng.choice("aabbc x1 _\n.,") for _ in range(n))


def py_leftmost_reports(pattern, text, flags=0):
    """All (from, to) with non-empty match text[from:to] of `pattern` in full context, leftmost from
    per to -- brute force with Python's backtracking `re` (lookahead pins the end offset)."""
    import re
    out = []
    n = len(text)
    for to in range(1, n + 1):
        rx = re.compile("(?:%s)(?=[\\s\\S]{%d}\\Z)" % (pattern, n - to), flags)
        for frm in range(0, to):
            m = rx.match(text, frm)
            if m and m.end() == to:
                out.append((frm, to))
                break
    return out
