#!/usr/bin/env python3
"""bench.py -- prints ONE JSON line (driver contract).

A "step" is one pass of the hot path over one batch of synthetic documents already resident in
HBM.  One process per GPU (torch.distributed / RCCL when launched by torchrun); documents shard
across ranks with no data-path collective (weak scaling: fixed work per GPU); the only collective
is the final reduce of the counters and the max over ranks of the wall time
(struspattern_amd/dist.py, the same code the world-2 gloo test runs).

Workloads (--workload):
  pipeline  (default) BASELINE.json configs[4] per-GPU shard: 10k regexes + sentence delimiter ->
            lexems stay in HBM -> 10k two-term token rules; value = GB/s of text scanned.
            Headline shape: 64 KiB documents; the same bytes cut into
            16 KiB documents are timed beside it (config.shapes) unless --docbytes fixes one shape.
  lexer     configs[1]: 256 regexes over 64 KiB ASCII documents, lexer kernel only
  l2        configs[2]: 10k rules over a pre-tokenized event stream, rule-automaton kernel only
  trees     configs[3]: depth-8 nested within/sequence expression trees (not flat: the general automaton kernel, state in HBM)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "GB/s input text scanned + matches/s, 10k-regex lexer + 10k-rule automaton"
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_bench_pmc_summary.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SPA_BENCH_WORKLOAD", "pipeline"))
    ap.add_argument("--regexes", type=int, default=0)
    ap.add_argument("--rules", type=int, default=10000)
    ap.add_argument("--docs", type=int, default=0)
    ap.add_argument("--docbytes", type=int, default=0, help="document size (default: 64 KiB headline + 16 KiB beside it in the pipeline workload, 64 KiB in the lexer workload)")
    ap.add_argument("--docsize", type=int, default=1000, help="tokens per document (workload l2)")
    ap.add_argument("--features", type=int, default=10000, help="distinct tokens (workload l2)")
    ap.add_argument("--op", default="", help="fix the rule operator (default: the 5-way Zipf mix)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target size of the bounded CPU-baseline sample")
    return ap.parse_args()


def size_until_ok(run, counters, status, reserve, grow, ndocs, what):
    """First passes size the output buffers / arenas: the kernels count past the capacity."""
    c = None
    for _ in range(10):
        run()
        c = counters()
        if c["failed_docs"] == 0:
            return c
        st = status(ndocs)
        codes = sorted(set(int(x) for x in st[st != 0]))
        if 9 in codes:
            reserve(c)
        if 2 in codes:
            grow()
        if not (set(codes) <= {2, 9}):
            raise SystemExit("bench: %s documents failed with status codes %s" % (what, codes))
    raise SystemExit("bench: %s: %d documents still failing" % (what, c["failed_docs"]))


class Shape:
    """One cut of the corpus into documents (device-resident offsets)."""

    def __init__(self, name, offs, torch):
        self.name = name
        self.offs = offs
        self.ndocs = len(offs) - 1
        self.d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        self.lex_out = None
        self.nlexems = 0


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from struspattern_amd import dist as spdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < 1:   # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py needs a GPU: the match path has no CPU fallback")
    if world > 1:
        # the process group comes first: nothing touches the GPU before the rendezvous
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import struspattern_amd as spa
    from struspattern_amd import synth

    wl = args.workload
    stream = torch.cuda.current_stream().cuda_stream
    lctx = mctx = None
    pats = rules = None
    text = lex = None
    nbytes = 0
    shapes = []
    h2d_ms = None
    if wl in ("pipeline", "lexer"):
        nreg = args.regexes or (10000 if wl == "pipeline" else 256)
        vocab = synth.vocabulary(30000, 1)
        if wl == "pipeline":
            pats, rules = synth.pipeline_workload(nreg, args.rules, vocab, seed=4)
        else:
            pats = synth.lexer_patterns(nreg, vocab, seed=1)
        # 805 MB of text per step.  The corpus is generated as 16 KiB pieces that end at word boundaries;
        # a 64 KiB document is four consecutive pieces, so both shapes scan the very same bytes.
        if wl == "pipeline" and not args.docbytes:
            piece, group = 16384, [("64KiB", 4), ("16KiB", 1)]
        else:
            piece = args.docbytes or 65536
            group = [("%dKiB" % (piece // 1024), 1)]
        npieces = args.docs * group[0][1] if args.docs else (805306368 // piece)
        text, offs = synth.text_documents(npieces, piece, vocab, seed=1000 + rank, utf8=(wl == "pipeline"))
        nbytes = len(text)
        lxi = spa.PatternLexerInstance()
        synth.apply_lexer_patterns(lxi, pats)
        lctx = lxi.createContext(local_rank)
        host_text = torch.frombuffer(bytearray(text), dtype=torch.uint8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d_text = host_text.cuda()
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) * 1e3     # pageable host memory -> HBM, reported separately (never part of `value`)
        for name, g in group:
            shapes.append(Shape(name, np.ascontiguousarray(offs[::g]) if (len(offs) - 1) % g == 0 else np.concatenate([offs[:-1:g], offs[-1:]]), torch))
    else:
        if wl == "trees":
            rules = synth.tree_rules(args.rules if args.rules != 10000 else 1000, 60, 8, seed=3)
            # (every token keys about 1900 of these programs, 130 x the installs per event of the pipeline's rule set: shorter documents)
            if args.docsize == 1000:
                args.docsize = 200
            lex, offs = synth.tree_documents(args.docs or 1024, args.docsize, 60, seed=1000 + rank)
        else:
            rules = synth.random_rules(args.rules, args.features, seed=2, op=(args.op or None))
            lex, offs = synth.random_documents(args.docs or 10000, args.docsize, args.features, seed=1000 + rank)
        d_lex = torch.from_numpy(lex.view(np.int32)).cuda()
        sh = Shape("%dtok" % args.docsize, offs, torch)
        sh.nlexems = len(lex)
        shapes.append(sh)
    if wl in ("pipeline", "l2", "trees"):
        mi = spa.PatternMatcherInstance()
        if wl == "trees":
            synth.apply_trees(mi, rules)
        else:
            synth.apply_rules(mi, rules)
        mctx = mi.createContext(local_rank)

    def run_lexer(sh):
        sh.lex_out = lctx.matchDocsDevice(d_text.data_ptr(), sh.d_offs.data_ptr(), sh.ndocs, nbytes, stream)

    def run_matcher(sh):
        if wl in ("l2", "trees"):
            mctx.matchDocsDevice(d_lex.data_ptr(), sh.d_offs.data_ptr(), sh.ndocs, sh.nlexems, stream)
        else:
            mctx.matchLexedDevice(sh.lex_out.d_lexems, sh.lex_out.d_doc_ranges, sh.ndocs, sh.nlexems, stream)

    def step(sh):
        if lctx is not None:
            run_lexer(sh)
        if mctx is not None:
            run_matcher(sh)

    # sizing passes (untimed): output buffers and per-document working sets grow to what the corpus needs
    for sh in shapes:
        if lctx is not None:
            lc = size_until_ok(lambda: run_lexer(sh), lctx.batchCounters, lctx.batchStatus,
                               lambda c: lctx.reserveOutput(int(c["lexems"] * 1.2) + 1024), lctx.growArena, sh.ndocs, "lexer")
            sh.nlexems = int(lc["lexems"])
        if mctx is not None:
            size_until_ok(lambda: run_matcher(sh), mctx.batchCounters, mctx.batchStatus,
                          lambda c: mctx.reserveOutput(int(c["results"] * 1.2) + 1024, int(c["items"] * 1.2) + 1024),
                          mctx.growArena, sh.ndocs, "matcher")

    def measure(sh, steps, warmup, collective):
        for _ in range(warmup):
            step(sh)
        torch.cuda.synchronize()
        if collective and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(sh)
        torch.cuda.synchronize()
        if collective and world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # per-kernel launch durations: HIP events recorded by the library on the launch stream
        l1_ms, l2_ms, scan_ms, words_ms, post_ms = [], [], [], [], []
        for _ in range(min(steps, 5)):
            step(sh)
            if lctx is not None:
                l1_ms.append(lctx.lastKernelMs())
                a, b, d = lctx.lastKernelMsSplit3()
                scan_ms.append(a)
                words_ms.append(b)
                post_ms.append(d)
            if mctx is not None:
                l2_ms.append(mctx.lastKernelMs())
        lcount = lctx.batchCounters() if lctx is not None else {"lexems": 0, "bytes": 0, "failed_docs": 0, "raw_reports": 0, "word_reports": 0}
        mcount = mctx.batchCounters() if mctx is not None else {"results": 0, "items": 0, "events": 0, "failed_docs": 0}
        if lcount["failed_docs"] or mcount["failed_docs"]:
            raise SystemExit("bench: documents failed in the timed region")
        return {"dt": dt, "steps": steps, "l1_ms": float(np.mean(l1_ms)) if l1_ms else 0.0, "l2_ms": float(np.mean(l2_ms)) if l2_ms else 0.0,
                "l1_scan_ms": float(np.mean(scan_ms)) if scan_ms else 0.0, "l1_words_ms": float(np.mean(words_ms)) if words_ms else 0.0, "l1_post_ms": float(np.mean(post_ms)) if post_ms else 0.0,
                "word_reports": int(lcount.get("word_reports", 0)), "l2_kernel": (mctx.kernelName() if mctx is not None else None),
                "scan_kernel": (lctx.scanKernelName() if lctx is not None and scan_ms else None),
                "words_kernel": (lctx.wordsKernelName() if lctx is not None and scan_ms else None),
                "raw_reports": int(lcount["raw_reports"]), "lexems": int(lcount["lexems"]), "events": int(mcount["events"]), "results": int(mcount["results"]), "items": int(mcount["items"])}

    # secondary shapes first (short, no barrier), the headline shape last so that its outputs are the
    # ones left in the device buffers for the parity sample
    secondary = {}
    for sh in shapes[1:]:
        secondary[sh.name] = measure(sh, max(2, min(args.steps, 5)), 1, False)
    head = shapes[0]
    m = measure(head, args.steps, args.warmup, True)

    # the path's only collective: integer counters summed, wall time maxed over the ranks
    local = {"bytes": nbytes, "lexems": m["lexems"], "events": m["events"], "results": m["results"]}
    tot, dt = spdist.reduce_step(local, m["dt"], device=torch.device("cuda", local_rank))

    if rank == 0:
        out = report(args, wl, world, pats, head, m, tot, dt, secondary, nbytes, h2d_ms)
        if not args.no_cpu_baseline:
            base, parity = cpu_baseline(args, wl, pats, rules, text, head, lex, lctx, mctx)
            ref = base.get("l1_reference") or {}
            if ref.get("GBps_allcores_extrapolated") and m["l1_ms"] > 0:
                base["gpu_vs_reference_cpu"] = (nbytes / (m["l1_ms"] * 1e-3) / 1e9) / ref["GBps_allcores_extrapolated"]
            out["cpu_baseline"] = base
            out["parity_sample"] = parity
        if world == 1 and mctx is not None and not args.no_cpu_baseline:
            if wl != "trees":
                out["canonical_order_mode"] = canonical_order_mode(wl, rules, head, d_lex if wl == "l2" else None, local_rank, m, nbytes)
        print(json.dumps(out))
        if out.get("parity_sample") and not out["parity_sample"]["ok"]:
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit("bench: GPU results differ from the oracle on the parity sample: %s" % out["parity_sample"].get("mismatch"))
    if world > 1:
        dist.destroy_process_group()


def roofline_of(kernel, kms, kbytes, traffic=None, source=None):
    achieved = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "kernel": kernel, "kernel_ms": kms, "algorithmic_bytes_per_launch": kbytes}
    if source:
        r["traffic_source"] = source
    return r


def report(args, wl, world, pats, head, m, tot, dt, secondary, nbytes, h2d_ms):
    steps = args.steps
    gbytes, gresults, gevents, glexems = tot["bytes"], tot["results"], tot["events"], tot["lexems"]
    if wl in ("l2", "trees"):
        value, unit, metric = gresults * steps / dt, "matches/s", METRIC + " [rule-automaton stage only: matches/s]"
    else:
        value, unit, metric = gbytes * steps / dt / 1e9, "GB/s", METRIC
    # algorithmic bytes per launch: L1 = text + 16 B x lexems; L2 = 16 B x events + 36 B x results;
    # pipeline = text + 36 B x results.  The lexer runs as two kernels with the raw reports (16 B each) between
    # them: scan = text + raw reports written; words (literals and word shapes, found where word runs end) = text + word reports
    # written; post = raw reports + word reports read + lexems written (it no longer reads the text, only the bytes of matches with symbols).
    b_l1 = float(nbytes) + 16.0 * m["lexems"]
    b_l2 = 16.0 * m["events"] + 36.0 * m["results"]
    roofs = {}
    scanname = m.get("scan_kernel") or "spa_l1_scan_kernel"
    wordsname = m.get("words_kernel") or "spa_l1_words_kernel"
    if m["l1_ms"] > 0:
        roofs[scanname] = roofline_of(scanname, m["l1_scan_ms"], float(nbytes) + 16.0 * m["raw_reports"],
                                      *pmc_traffic(scanname, wl, args, m["l1_scan_ms"]))
        if m["l1_words_ms"] > 0.05:
            roofs[wordsname] = roofline_of(wordsname, m["l1_words_ms"], float(nbytes) + 16.0 * m["word_reports"],
                                           *pmc_traffic(wordsname, wl, args, m["l1_words_ms"]))
            post_bytes = 16.0 * (m["raw_reports"] + m["word_reports"]) + 16.0 * m["lexems"]
        else:
            post_bytes = float(nbytes) + 16.0 * m["raw_reports"] + 16.0 * m["lexems"]
        roofs["spa_l1_post_kernel"] = roofline_of("spa_l1_post_kernel", m["l1_post_ms"], post_bytes,
                                                  *pmc_traffic("spa_l1_post_kernel", wl, args, m["l1_post_ms"]))
    l2name = m.get("l2_kernel") or "spa_l2_match_kernel"
    if m["l2_ms"] > 0:
        roofs[l2name] = roofline_of(l2name, m["l2_ms"], b_l2, *pmc_traffic(l2name, wl, args, m["l2_ms"]))
    dominant = max(roofs.values(), key=lambda r: r["kernel_ms"])
    if m["l1_ms"] > 0:
        roofs["lexer"] = roofline_of("scan + post", m["l1_ms"], b_l1)
    if wl == "pipeline":
        roofs["pipeline"] = roofline_of("lexer + automaton", m["l1_ms"] + m["l2_ms"], float(nbytes) + 36.0 * m["results"])
    workload = {
        "pipeline": "configs[4] per-GPU shard: %d regexes + %d token rules, %d docs x %s UTF-8 per step (the same %d MB resident in HBM every step; configs[4] asks 32 GB per GPU: %d such steps)" % (
            len(pats) if pats else 0, args.rules, head.ndocs, head.name, nbytes // 1000000, max(1, int(32e9 // max(1, nbytes)))),
        "lexer": "configs[1]: %d regexes, %d docs x %s ASCII per step (lexer only)" % (len(pats) if pats else 0, head.ndocs, head.name),
        "l2": "configs[2]: %d rules (%s), %d docs x %d tokens per step (rule automaton only)" % (args.rules, args.op or "5-op Zipf mix", head.ndocs, args.docsize),
        "trees": "configs[3]: %d depth-8 expression trees over 60 features, %d docs x %d tokens per step (rule automaton only, general kernel)" % (args.rules if args.rules != 10000 else 1000, head.ndocs, args.docsize),
    }[wl]
    out = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64" if wl in ("pipeline", "lexer") else "u32", "data": "synthetic",
        "config": {"workload": workload, "bytes_per_step_per_gpu": nbytes, "lexems_per_step_per_gpu": m["lexems"],
                   "events_per_step_per_gpu": m["events"], "matches_per_step_per_gpu": m["results"]},
        "matches_per_s": gresults * steps / dt,
        "events_per_s": gevents * steps / dt,
        "lexems_per_s": glexems * steps / dt,
        "kernel_ms": {scanname: m["l1_scan_ms"], wordsname: m["l1_words_ms"], "spa_l1_post_kernel": m["l1_post_ms"], l2name: m["l2_ms"]},
        "roofline": dominant,
        "roofline_all": roofs,
    }
    if secondary:
        out["config"]["shapes"] = {head.name: {"GB/s": value / world if wl in ("pipeline", "lexer") else None, "ms_per_step": dt / steps * 1e3,
                                               "kernel_ms": [m["l1_ms"], m["l2_ms"]], "matches": m["results"]}}
        for name, s in secondary.items():
            out["config"]["shapes"][name] = {"GB/s": nbytes * s["steps"] / s["dt"] / 1e9, "ms_per_step": s["dt"] / s["steps"] * 1e3,
                                             "kernel_ms": [s["l1_ms"], s["l2_ms"]], "matches": s["results"],
                                             "note": "rank 0, same bytes cut into %s documents, %d steps, no barrier" % (name, s["steps"])}
    if h2d_ms is not None:
        out["h2d_staging_ms"] = h2d_ms
        out["h2d_inclusive_GBps"] = nbytes / (dt / steps + h2d_ms * 1e-3) / 1e9
    return out


def pmc_traffic(kernel_prefix, wl, args, kernel_ms):
    """HBM-side bytes per launch (read + write) of the kernel from the committed rocprofv3 PMC passes of this
    same command (tests/micro/profile_bench.sh -> profiles/r02_bench_pmc_summary.json; FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler on itself: the figure is
    reported only for the default configuration and only while the profiled kernel ran as long as the one
    timed here (within 15 %), i.e. it is the same kernel revision on the same work."""
    if wl != "pipeline" or args.docs or args.regexes or args.docbytes not in (0, 65536) or args.rules != 10000:
        return None, None
    try:
        with open(PMC_SUMMARY) as f:
            summ = json.load(f)
        # (several instances of a kernel are launched, the ones a batch does not need leave at once: the one that ran is the longest)
        cands = [k for name, k in summ["kernels"].items() if name.startswith(kernel_prefix)]
        if cands:
            k = max(cands, key=lambda k: float(k.get("kernel_ms_steady", k.get("kernel_ms", 0.0))))
            pms = float(k.get("kernel_ms_steady", k.get("kernel_ms", 0.0)))
            if pms <= 0 or abs(pms - kernel_ms) > 0.15 * kernel_ms:
                return None, None
            return (float(k["hbm_read_bytes_per_launch"]) + float(k["hbm_write_bytes_per_launch"]),
                    "profiles/" + os.path.basename(PMC_SUMMARY) + " (%s)" % summ.get("commit", "?"))
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def canonical_order_mode(wl, rules, head, d_lex, local_rank, m, nbytes):
    """NOT part of `value`: the rule stage of the same step on the opt-in join prototype (csrc/l2_join.h, SPA_L2_JOIN=1), which
    evaluates the rules' meaning without materialising rule instances: the oracle's result SETS for rule sets that were not
    compile()d (tests/test_l2_join_gpu.py), not the reference's order of the results inside a document, and not the optimizer's
    alternative keys (the counts of the two engines are reported side by side).  Reported beside the exact engine's time."""
    try:
        import torch
        import struspattern_amd as spa
        from struspattern_amd import synth
        os.environ["SPA_L2_JOIN"] = "1"
        mi = spa.PatternMatcherInstance()
        synth.apply_rules(mi, rules)
        jctx = mi.createContext(local_rank)
        os.environ.pop("SPA_L2_JOIN", None)
        if jctx.kernelKind() != 2:
            return {"available": False}
        stream = torch.cuda.current_stream().cuda_stream

        def run():
            if wl == "l2":
                jctx.matchDocsDevice(d_lex.data_ptr(), head.d_offs.data_ptr(), head.ndocs, head.nlexems, stream)
            else:
                jctx.matchLexedDevice(head.lex_out.d_lexems, head.lex_out.d_doc_ranges, head.ndocs, head.nlexems, stream)
        c = size_until_ok(run, jctx.batchCounters, jctx.batchStatus,
                          lambda c: jctx.reserveOutput(int(c["results"] * 1.2) + 1024, 1024), jctx.growArena, head.ndocs, "join prototype")
        ms = []
        for _ in range(3):
            run()
            ms.append(jctx.lastKernelMs())
        l2 = float(np.mean(ms))
        out = {"available": True, "l2_ms": l2, "events_per_s": c["events"] / (l2 * 1e-3), "results": int(c["results"]),
               "results_of_the_exact_engine": m["results"], "note": "result sets only, order inside a document not the reference's; pinned against the oracle for rule sets WITHOUT compile() only -- "
                       "the optimizer's alternative keys are not modelled, hence the two counts differ slightly; never part of `value`"}
        if wl == "pipeline":
            out["pipeline_GBps_with_it"] = nbytes / ((m["l1_ms"] + l2) * 1e-3) / 1e9
        return out
    except (Exception, SystemExit) as e:          # the line must not depend on the prototype
        os.environ.pop("SPA_L2_JOIN", None)
        return {"available": False, "error": str(e)[:200]}


def hyperscan_reference(pats, sample_text, sample_offs, ncores):
    """The reference's own CPU lexer stage is Intel Hyperscan (src/patternLexer.cpp:875-879).  Probe for libhs; when it is there, time
    hs_scan over the same sample with the reference's compile flags (src/patternLexer.cpp:391-405: the options | HS_FLAG_SOM_LEFTMOST |
    HS_FLAG_UTF8, mode HS_MODE_BLOCK, :1078-1086) -- one thread, a callback that only counts, i.e. an upper bound of what the reference's
    handler would let through.  Returns the `l1_reference` object of cpu_baseline."""
    import ctypes
    import ctypes.util
    name = ctypes.util.find_library("hs")
    if not name:
        for cand in ("libhs.so.5", "libhs.so.4", "libhs.so"):
            try:
                ctypes.CDLL(cand)
                name = cand
                break
            except OSError:
                pass
    if not name:
        return {"probed": True, "found": False, "note": "libhs (Intel Hyperscan) is not installed on this box: the reference's lexer stage cannot be timed here"}
    try:
        hs = ctypes.CDLL(name)
        HS_FLAG_DOTALL, HS_FLAG_UTF8, HS_FLAG_SOM_LEFTMOST, HS_MODE_BLOCK = 2, 32, 256, 1
        n = len(pats)
        exprs = (ctypes.c_char_p * n)(*[e.encode() for _, e, _, _, _ in pats])
        flags = (ctypes.c_uint * n)(*([HS_FLAG_DOTALL | HS_FLAG_UTF8 | HS_FLAG_SOM_LEFTMOST] * n))
        ids = (ctypes.c_uint * n)(*range(1, n + 1))
        db, err, scratch = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = hs.hs_compile_multi(exprs, flags, ids, n, HS_MODE_BLOCK, None, ctypes.byref(db), ctypes.byref(err))
        tc = time.perf_counter() - t0
        if rc != 0:
            return {"probed": True, "found": True, "library": name, "error": "hs_compile_multi returned %d" % rc}
        hs.hs_alloc_scratch(db, ctypes.byref(scratch))
        count = [0]
        CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_void_p)

        def on_match(i, frm, to, fl, ctx):
            count[0] += 1
            return 0
        cb = CB(on_match)
        t0 = time.perf_counter()
        for d in range(len(sample_offs) - 1):
            doc = sample_text[int(sample_offs[d]):int(sample_offs[d + 1])]
            hs.hs_scan(db, doc, len(doc), 0, scratch, cb, None)
        dt = time.perf_counter() - t0
        return {"probed": True, "found": True, "library": name, "GBps_1thread": len(sample_text) / dt / 1e9, "raw_matches": count[0], "compile_s": tc,
                "note": "hs_scan with the reference's flags on the oracle sample, 1 thread, counting callback through ctypes (callback overhead included)"}
    except Exception as e:          # a library of another ABI must not take the bench line down
        return {"probed": True, "found": True, "library": name, "error": str(e)[:200]}


def first_difference(gl, gr, rl, rr):
    """compares lexems / results / items / statistics of the GPU sample with the oracle's; None when equal"""
    if not (np.array_equal(gl.doc_offsets, rl[1]) and np.array_equal(gl.lexems, rl[0])):
        return "lexems"
    if gr is None:
        return None
    if not np.array_equal(gr.doc_offsets, rr.doc_offsets):
        return "result counts per document"
    if not np.array_equal(gr.results[:, :7], rr.results[:, :7]):
        return "result tuples"
    if not (np.array_equal(gr.results[:, 8], rr.results[:, 8]) and np.array_equal(gr.items, rr.items)):
        return "result items"
    if not np.array_equal(gr.stats, rr.stats):
        return "statistics"
    return None


def cpu_baseline(args, wl, pats, rules, text, head, lex, lctx, mctx):
    """The oracle (CPU restatement of the reference path) timed on a bounded sample of rank 0's shard, and
    its outputs compared with what the GPU produced for the same documents in the last timed launch.
    The reference's own lexer stage is Intel Hyperscan, which is not available here: the L1 number is
    the scalar NFA restatement ("port"), not Hyperscan.  SURVEY.md 8(d): L2 at 1 thread and at all cores."""
    import oracle
    from struspattern_amd import synth
    # a one-GPU box grants about 16 host cores to the job whatever os.cpu_count() says
    ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    offs = head.offs
    if wl in ("l2", "trees"):
        o = oracle.L2Matcher()
        if wl == "trees":
            synth.apply_trees(o, rules)
        else:
            synth.apply_rules(o, rules)
        nd = max(1, min(len(offs) - 1, int(args.cpu_seconds * 100000 / max(1, args.docsize))))
        sub = synth.lexems5(lex[:int(offs[nd])])
        t0 = time.perf_counter()
        r = o.run(sub, offs[:nd + 1], nthreads=1)
        t1 = time.perf_counter()
        rall = o.run(sub, offs[:nd + 1], nthreads=ncores)
        t2 = time.perf_counter()
        gr = mctx.batchFetch(0, nd)
        diff = None
        for what, a, b in (("result counts per document", gr.doc_offsets, r.doc_offsets), ("result tuples", gr.results[:, :7], r.results[:, :7]),
                           ("result items", gr.items, r.items), ("statistics", gr.stats, r.stats)):
            if diff is None and not np.array_equal(a, b):
                diff = what
        base = {"value": len(rall.results) / (t2 - t1), "unit": "matches/s", "cores": ncores, "kind": "port",
                "l2": {"events_per_s_1thread": int(offs[nd]) / (t1 - t0), "events_per_s_allcores": int(offs[nd]) / (t2 - t1),
                       "matches_per_s_1thread": len(r.results) / (t1 - t0), "cores": ncores},
                "sample": "first %d documents of rank 0's shard (%d events), oracle/l2_oracle.cpp, 1 thread %.1f s, %d threads %.1f s" % (nd, int(offs[nd]), t1 - t0, ncores, t2 - t1)}
        return base, {"docs": nd, "ok": diff is None, "mismatch": diff, "compared": "results in firing order, items, statistics"}
    ol = oracle.L1Lexer()
    synth.apply_lexer_patterns(ol, pats)
    # calibrate on one 16 KB slice, then size the sample for about cpu_seconds of wall time on all cores
    cal = min(int(offs[1]), 16384)
    t0 = time.perf_counter()
    ol.matchDocs(text[:cal], np.array([0, cal], np.uint64), nthreads=1)
    t_byte = max(1e-9, (time.perf_counter() - t0) / max(1, cal))
    doc_bytes = int(offs[1]) if len(offs) > 1 else 1
    nd = max(1, min(len(offs) - 1, int(args.cpu_seconds * ncores / (t_byte * doc_bytes) / 1.5)))
    nthreads = min(ncores, nd)
    sub_text = text[:int(offs[nd])]
    t0 = time.perf_counter()
    lexems, loffs = ol.matchDocs(sub_text, offs[:nd + 1], nthreads=nthreads)
    t1 = time.perf_counter()
    ref = hyperscan_reference(pats, sub_text, offs[:nd + 1], ncores)
    base = {"value": len(sub_text) / (t1 - t0) / 1e9, "unit": "GB/s", "cores": nthreads, "kind": "port",
            "l1": {"GBps": len(sub_text) / (t1 - t0) / 1e9, "cores": nthreads, "kind": "port",
                   "note": "oracle lexer: scalar NFA restatement of Hyperscan's report semantics, NOT Hyperscan -- no yardstick for the GPU lexer"},
            "l1_reference": ref,
            "gpu_vs_reference_cpu": None,
            "gpu_vs_reference_cpu_note": ("GPU lexer GB/s / (libhs GB/s per thread x %d host cores) is in cpu_baseline.l1_reference" % ncores) if ref.get("GBps_1thread")
                                         else "null: the reference's CPU lexer (Hyperscan) could not be timed on this box (%s)" % (ref.get("note") or ref.get("error") or "probe failed"),
            "sample": "first %d documents of rank 0's shard (%d bytes), oracle lexer %d threads %.1f s" % (nd, len(sub_text), nthreads, t1 - t0)}
    gl = lctx.batchFetch(0, nd)
    gr = rr = None
    if wl == "pipeline":
        om = oracle.L2Matcher()
        synth.apply_rules(om, rules)
        rr = om.run(synth.lexems5(lexems), loffs, nthreads=nthreads)
        t2 = time.perf_counter()
        base["value"] = len(sub_text) / (t2 - t0) / 1e9
        base["matches_per_s"] = len(rr.results) / (t2 - t0)
        base["sample"] += " + oracle automaton %.2f s" % (t2 - t1)
        gr = mctx.batchFetch(0, nd)
        # L2 restatement alone: a larger slice of the GPU lexer's output (equal to the oracle
        # lexer's on the parity sample) so that the timing is not dominated by thread start-up
        n2 = max(nd, min(len(offs) - 1, int(args.cpu_seconds * 250000 * 6 / max(1, doc_bytes))))
        g2 = lctx.batchFetch(0, n2)
        l5 = synth.lexems5(g2.lexems)
        t3 = time.perf_counter()
        r1 = om.run(l5, g2.doc_offsets, nthreads=1)
        t4 = time.perf_counter()
        om.run(l5, g2.doc_offsets, nthreads=ncores)
        t5 = time.perf_counter()
        base["l2"] = {"events_per_s_1thread": len(l5) / (t4 - t3), "events_per_s_allcores": len(l5) / (t5 - t4),
                      "matches_per_s_1thread": len(r1.results) / (t4 - t3), "cores": ncores, "kind": "port",
                      "sample": "%d documents (%d events): lexer output of the GPU for rank 0's first documents, oracle/l2_oracle.cpp" % (n2, len(l5))}
    if ref.get("GBps_1thread"):
        ref["GBps_allcores_extrapolated"] = ref["GBps_1thread"] * ncores
    diff = first_difference(gl, gr, (lexems, loffs), rr)
    return base, {"docs": nd, "ok": diff is None, "mismatch": diff,
                  "compared": "lexems, results in firing order, items, statistics of the last timed launch vs the oracle"}


if __name__ == "__main__":
    main()
