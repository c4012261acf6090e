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
