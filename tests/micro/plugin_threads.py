"""Plugin path under the reference's threading model (one Context per thread over a shared Instance): documents/s and MB/s of
struspattern_amd/_build/testStrusInterface --threads N at several thread counts (steady state: a first pass sizes the buffers).
usage: python tests/micro/plugin_threads.py [ndocs] [docbytes] [threads,threads,..]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from struspattern_amd import build as spbuild, synth

ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
docbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
threads = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8,16").split(",")]
lib, module, testbin = spbuild.build_host()
vocab = synth.vocabulary(30000, 1)
pats, rules = synth.pipeline_workload(1000, 1000, vocab, 4)
text, offs = synth.text_documents(ndocs, docbytes, vocab, 1000, utf8=True)
fixture = os.path.join(tempfile.mkdtemp(), "threads.txt")
with open(fixture, "w", encoding="utf8") as f:
    f.write("OPTION\tDOTALL\n")
    for lid, expr, residx, level, posbind in pats:
        f.write("LEXEM\t%d\t%s\t%d\t%d\t%d\n" % (lid, expr, residx, level, 1 if posbind == "content" else 0))
    for name, op, rg, params in rules:
        delim = synth.DELIM if op in ("sequence_struct", "within_struct") else 0
        f.write("XRULE\t%s\t%s\t%d\t%d\t%d\t%s\n" % (name, op, rg, delim, len(params), "\t".join("%d\tA%d" % (t, i) for i, t in enumerate(params))))
    for d in range(len(offs) - 1):
        f.write("DOC\t%s\n" % text[int(offs[d]):int(offs[d + 1])].hex())
for n in threads:
    p = subprocess.run([testbin, "--threads", str(n), fixture, "4"], capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        print("threads %d: FAILED %s" % (n, p.stderr.strip()[-300:]))
        continue
    for ln in p.stdout.splitlines():
        f = ln.split("\t")
        if f[0] == "THREADS":
            nd, nb, secs = int(f[3]), int(f[5]), float(f[7])
            print("plugin path: %2d threads, %d x %d B documents, 1000 regexes + 1000 rules: %.0f documents/s, %.2f MB/s (%.2f ms per document and thread)" % (
                n, ndocs, docbytes, nd / secs, nb / secs / 1e6, secs * n / nd * 1e3), flush=True)
