"""The C++ host side of the drop-in (struspattern_amd/host): builds on CPU, exports the reference's
factory functions and module entry point; on a GPU the C++ ports of the reference's two known-answer
tests run through the strus plugin interfaces and through dlopen + `entryPoint`."""
import ctypes
import os
import subprocess

import pytest

from struspattern_amd import build as spbuild
from tests import l1_cases, l2_cases


def _built():
    return spbuild.build_host()


def test_module_exports_reference_symbols():
    lib, module, testbin = _built()
    m = ctypes.CDLL(module)
    assert hasattr(m, "entryPoint")                      # src/modstrus_analyzer_pattern.cpp:59-61
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    assert "createPatternLexer_std" in out and "createPatternMatcher_std" in out   # include/strus/lib/pattern.hpp:27-32


def test_compile_option_names_equal_the_references():
    """getCompileOptionNames of both interfaces, C++ shim and Python mirror, against the reference's lists
    (tests/golden/option_names.json <- src/patternLexer.cpp:1154-1163, src/patternMatcher.cpp:707-716)."""
    import json
    import struspattern_amd as spa
    with open(os.path.join(os.path.dirname(__file__), "golden", "option_names.json")) as f:
        want = json.load(f)
    assert spa.PatternLexer().getCompileOptionNames() == want["lexer"]
    assert spa.PatternMatcher().getCompileOptionNames() == want["matcher"]
    lib, module, testbin = _built()
    out = subprocess.check_output([testbin, "--options"], text=True)
    got = {"lexer": [], "matcher": []}
    for ln in out.splitlines():
        k, v = ln.split("\t")
        got[k].append(v)
    assert got["lexer"] == want["lexer"] and got["matcher"] == want["matcher"]


def _write_fixtures(tmp, regex_case=0):
    case = l2_cases.load("simple_token_pattern_match.json")
    simple = os.path.join(tmp, "simple.txt")
    with open(simple, "w") as f:
        for r in case["rules"]:
            f.write("RULE\t%s\t%d\t%s\t%d\t%s\t%d\n" % (r["name"], r["terms"][0][0], r["terms"][0][1], r["terms"][1][0], r["terms"][1][1], r["range"]))
            for p in r["expected_ordpos"]:
                f.write("EXPECT\t%s\t%d\n" % (r["name"], p))
        for lx in l2_cases.simple_doc(case):
            f.write("DOC\t%d\t%d\t%d\n" % (lx[0], lx[1], lx[3]))
    rc = l1_cases.load_char_regex_cases()[regex_case]
    regex = os.path.join(tmp, "regex.txt")
    with open(regex, "w", encoding="utf8") as f:
        f.write("OPTION\tDOTALL\n")
        for pid, expr, residx, level, haspos in rc["patterns"]:
            f.write("LEXEM\t%d\t%s\t%d\t%d\t%d\n" % (pid, expr, residx, level, int(haspos)))
        for symid, patid, name in rc["symbols"]:
            f.write("SYMBOL\t%d\t%d\t%s\n" % (symid, patid, name))
        f.write("SRC\t%s\n" % rc["src"])
        for e in rc["result"]:
            f.write("EXPECT\t%d\t%d\t%d\t%d\n" % tuple(e))
    return simple, regex


@pytest.mark.gpu
@pytest.mark.parametrize("regex_case", [0, 1, 2])     # charRegexMatch case 1 (36 lexems), cases 2 and 3 (`abc ~1`, ASCII and UTF-8)
def test_reference_known_answer_tests_through_the_cpp_interfaces(tmp_path, regex_case):
    lib, module, testbin = _built()
    simple, regex = _write_fixtures(str(tmp_path), regex_case)
    p = subprocess.run([testbin, simple, regex, module], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == "OK"
    assert "simpleTokenPatternMatch OK" in p.stderr and "charRegexMatch OK" in p.stderr and "module entryPoint OK" in p.stderr


def _fnv(h, b):
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _fnv_u(h, v):
    return _fnv(h, int(v).to_bytes(8, "little"))


def _fnv_s(h, s):
    return _fnv(h, s.encode() + b"\0")


@pytest.mark.gpu
def test_plugin_path_with_eight_threads_equals_the_oracle(tmp_path):
    """The reference's threading model through the C++ plugin interfaces: one Context per thread over a shared Instance
    (tests/randomTokenPatternMatch/src/testRandomTokenPatternMatch.cpp:325-345).  Eight threads, each with its own lexer and matcher
    context, run every document (match -> putInput per lexem -> fetchResults); every thread must get, per document, the results
    the oracle gets (names, positions, items: compared through a hash of the ordered result list).  Contexts have their own
    streams and are dealt over the visible devices (SPA_DEVICE pins them), so the threads overlap."""
    import numpy as np
    import oracle
    import struspattern_amd as spa
    from struspattern_amd import synth
    lib, module, testbin = _built()
    vocab = synth.vocabulary(2000, 5)
    pats, rules = synth.pipeline_workload(300, 400, vocab, 11)
    text, offs = synth.text_documents(24, 3000, vocab, 12, utf8=True)
    fixture = os.path.join(str(tmp_path), "threads.txt")
    with open(fixture, "w", encoding="utf8") as f:
        f.write("OPTION\tDOTALL\n")
        for lid, expr, residx, level, posbind in pats:
            f.write("LEXEM\t%d\t%s\t%d\t%d\t%d\n" % (lid, expr, residx, level, 1 if posbind == "content" else 0))
        for name, op, rg, params in rules:
            delim = synth.DELIM if op in ("sequence_struct", "within_struct") else 0
            f.write("XRULE\t%s\t%s\t%d\t%d\t%d\t%s\n" % (name, op, rg, delim, len(params), "\t".join("%d\tA%d" % (t, i) for i, t in enumerate(params))))
        for d in range(len(offs) - 1):
            f.write("DOC\t%s\n" % text[int(offs[d]):int(offs[d + 1])].hex())
    p = subprocess.run([testbin, "--threads", "8", fixture, "3"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    got = {}
    rate = None
    for ln in p.stdout.splitlines():
        f = ln.split("\t")
        if f[0] == "DOCSUM":
            got[int(f[1])] = (int(f[2]), int(f[3]))
        elif f[0] == "THREADS":
            rate = (int(f[1]), int(f[3]), int(f[5]), float(f[7]))
    assert rate and rate[0] == 8 and len(got) == len(offs) - 1
    # the oracle on the same documents
    ol = oracle.L1Lexer()
    synth.apply_lexer_patterns(ol, pats)
    om = oracle.L2Matcher()
    synth.apply_rules(om, rules)
    mi = spa.PatternMatcherInstance()
    synth.apply_rules(mi, rules)
    lex, loffs = ol.matchDocs(text, offs, nthreads=4)
    ref = om.run(synth.lexems5(lex), loffs)
    nonempty = 0
    for d in range(len(offs) - 1):
        h = 1469598103934665603
        r0, r1 = int(ref.doc_offsets[d]), int(ref.doc_offsets[d + 1])
        for r in ref.results[r0:r1]:
            h = _fnv_s(h, mi.patternName(int(r[0])))
            for v in (r[1], r[2], r[4], r[6]):
                h = _fnv_u(h, v)
            for it in ref.items[int(r[7]):int(r[7]) + int(r[8])]:
                h = _fnv_s(h, mi.variableName(int(it[0])))
                for v in (it[1], it[2], it[4], it[6]):
                    h = _fnv_u(h, v)
        assert got[d] == (r1 - r0, h), "document %d" % d
        nonempty += 1 if r1 > r0 else 0
    assert nonempty > len(offs) // 2
    print("plugin path, 8 threads: %d documents (%d bytes) in %.2f s = %.0f documents/s, %.2f MB/s" % (rate[1], rate[2], rate[3], rate[1] / rate[3], rate[2] / rate[3] / 1e6))
