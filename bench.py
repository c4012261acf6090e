#!/usr/bin/env python3
"""bench.py -- prints ONE JSON line (driver contract).

A "step" is one pass of the hot path over one batch of synthetic documents already resident in
HBM.  One process per GPU (torch.distributed / RCCL when launched by torchrun); documents shard
across ranks with no data-path collective (weak scaling: fixed work per GPU); the only collective
is the final reduce of the counters and the max over ranks of the wall time.

Workloads (--workload):
  pipeline  (default) BASELINE.json configs[4] per-GPU shard: 10k regexes + sentence delimiter ->
            lexems stay in HBM -> 10k two-term token rules; value = GB/s of text scanned
  lexer     configs[1]: 256 regexes over 64 KiB ASCII documents, lexer kernel only
  l2        configs[2]: 10k rules over a pre-tokenized event stream, rule-automaton kernel only
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SPA_BENCH_WORKLOAD", "pipeline"))
    ap.add_argument("--regexes", type=int, default=0)
    ap.add_argument("--rules", type=int, default=10000)
    ap.add_argument("--docs", type=int, default=0)
    ap.add_argument("--docbytes", type=int, default=0, help="document size (default: 16 KiB in the pipeline workload, 64 KiB in the lexer workload)")
    ap.add_argument("--docsize", type=int, default=1000, help="tokens per document (workload l2)")
    ap.add_argument("--features", type=int, default=10000, help="distinct tokens (workload l2)")
    ap.add_argument("--op", default="", help="fix the rule operator (default: the 5-way Zipf mix)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target size of the bounded CPU-baseline sample")
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


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the match path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import struspattern_amd as spa
    from struspattern_amd import synth

    wl = args.workload
    if not args.docbytes:
        args.docbytes = 16384 if wl == "pipeline" else 65536
    stream = torch.cuda.current_stream().cuda_stream
    lctx = mctx = None
    pats = rules = None
    text = offs = lex = None
    nbytes = 0
    if wl in ("pipeline", "lexer"):
        nreg = args.regexes or (10000 if wl == "pipeline" else 256)
        # documents per step: 805 MB of text as 16 KiB documents (a document is one sequential job of one
        # wave in both kernels: the shorter the jobs, the smaller the idle tail of a launch); the count is a
        # multiple of the resident waves of both kernels (4096 lexer / 3072 automaton slots on 256 CUs)
        ndocs = args.docs or (49152 if wl == "pipeline" else 8192)
        vocab = synth.vocabulary(30000, 1)
        if wl == "pipeline":
            pats, rules = synth.pipeline_workload(nreg, args.rules, vocab, seed=4)
        else:
            pats = synth.lexer_patterns(nreg, vocab, seed=1)
        text, offs = synth.text_documents(ndocs, args.docbytes, vocab, seed=1000 + rank, utf8=(wl == "pipeline"))
        nbytes = len(text)
        lxi = spa.PatternLexerInstance()
        synth.apply_lexer_patterns(lxi, pats)
        lctx = lxi.createContext(local_rank)
        d_text = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
        lex_out = {}

        def run_lexer():
            lex_out["o"] = lctx.matchDocsDevice(d_text.data_ptr(), d_offs.data_ptr(), ndocs, nbytes, stream)
        lc = size_until_ok(run_lexer, lctx.batchCounters, lctx.batchStatus,
                           lambda c: lctx.reserveOutput(int(c["lexems"] * 1.2) + 1024), lctx.growArena, ndocs, "lexer")
        nlexems = int(lc["lexems"])
    if wl in ("pipeline", "l2"):
        if wl == "l2":
            ndocs = args.docs or 10000
            rules = synth.random_rules(args.rules, args.features, seed=2, op=(args.op or None))
            lex, offs = synth.random_documents(ndocs, args.docsize, args.features, seed=1000 + rank)
            d_lex = torch.from_numpy(lex.view(np.int32)).cuda()
            d_loffs = torch.from_numpy(offs.view(np.int64)).cuda()
            nlexems = len(lex)
        mi = spa.PatternMatcherInstance()
        synth.apply_rules(mi, rules)
        mctx = mi.createContext(local_rank)

        def run_matcher():
            if wl == "l2":
                mctx.matchDocsDevice(d_lex.data_ptr(), d_loffs.data_ptr(), ndocs, nlexems, stream)
            else:
                o = lex_out["o"]
                mctx.matchLexedDevice(o.d_lexems, o.d_doc_ranges, ndocs, nlexems, stream)
        if os.environ.get("SPA_BENCH_EXTRA_GROW"):      # experiment: a working-set arena larger than needed (locality / TLB reach)
            for _ in range(int(os.environ["SPA_BENCH_EXTRA_GROW"])):
                mctx.growArena()
        mc = size_until_ok(run_matcher, mctx.batchCounters, mctx.batchStatus,
                           lambda c: mctx.reserveOutput(int(c["results"] * 1.2) + 1024, int(c["items"] * 1.2) + 1024),
                           mctx.growArena, ndocs, "matcher")

    def step():
        if lctx is not None:
            run_lexer()
        if mctx is not None:
            run_matcher()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # per-kernel launch durations: HIP events recorded by the library on the launch stream
    l1_ms, l2_ms = [], []
    for _ in range(args.steps):
        step()
        if lctx is not None:
            l1_ms.append(lctx.lastKernelMs())
        if mctx is not None:
            l2_ms.append(mctx.lastKernelMs())
    l1_ms = float(np.mean(l1_ms)) if l1_ms else 0.0
    l2_ms = float(np.mean(l2_ms)) if l2_ms else 0.0
    lcount = lctx.batchCounters() if lctx is not None else {"lexems": 0, "bytes": 0, "failed_docs": 0}
    mcount = mctx.batchCounters() if mctx is not None else {"results": 0, "items": 0, "events": 0, "failed_docs": 0}
    if lcount["failed_docs"] or mcount["failed_docs"]:
        raise SystemExit("bench: documents failed in the timed region")

    tot = torch.tensor([float(nbytes), float(lcount["lexems"]), float(mcount["events"]), float(mcount["results"])], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)   # the path's only collective: counters
        dt = float(tmax[0])
    gbytes, glexems, gevents, gresults = (float(x) for x in tot)

    if rank == 0:
        if wl == "l2":
            value, unit, metric = gresults * args.steps / dt, "matches/s", METRIC + " [rule-automaton stage only: matches/s]"
        else:
            value, unit, metric = gbytes * args.steps / dt / 1e9, "GB/s", METRIC
        # roofline of the dominant kernel (SURVEY.md 8(d) algorithmic bytes per launch)
        if l1_ms >= l2_ms:
            kname, kms = "spa_l1_lex_kernel", l1_ms
            kbytes = float(nbytes) + 16.0 * lcount["lexems"]
        else:
            kname, kms = "spa_l2_match_kernel", l2_ms
            kbytes = 16.0 * mcount["events"] + 36.0 * mcount["results"]
        achieved = kbytes / (kms * 1e-3) / 1e9
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64" if wl != "l2" else "u32", "data": "synthetic",
            "config": {"workload": {
                "pipeline": "configs[4] per-GPU shard: %d regexes + %d token rules, %d docs x %d B UTF-8 per step" % (len(pats) if pats else 0, args.rules, ndocs, args.docbytes),
                "lexer": "configs[1]: %d regexes, %d docs x %d B ASCII per step (lexer only)" % (len(pats) if pats else 0, ndocs, args.docbytes),
                "l2": "configs[2]: %d rules (%s), %d docs x %d tokens per step (rule automaton only)" % (args.rules, args.op or "5-op Zipf mix", ndocs, args.docsize),
            }[wl], "bytes_per_step_per_gpu": nbytes, "lexems_per_step_per_gpu": int(lcount["lexems"]),
                "events_per_step_per_gpu": int(mcount["events"]), "matches_per_step_per_gpu": int(mcount["results"])},
            "matches_per_s": gresults * args.steps / dt,
            "events_per_s": gevents * args.steps / dt,
            "lexems_per_s": glexems * args.steps / dt,
            "kernel_ms": {"spa_l1_lex_kernel": l1_ms, "spa_l2_match_kernel": l2_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(kname, wl, args), "kernel": kname, "kernel_ms": kms,
                         "algorithmic_bytes_per_launch": kbytes},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, wl, pats, rules, text, offs, lex, value, unit)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel, wl, args):
    """HBM-side bytes per launch of `kernel` (read + write) from the committed rocprofv3 PMC passes of
    this same command (tests/micro/profile_bench.sh -> profiles/r01_bench_pmc_summary.json; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler on itself,
    so the figure is only reported for the default workload and configuration it was collected on."""
    if wl != "pipeline" or args.docs or args.regexes:
        return None
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_summary.json")
    try:
        with open(path) as f:
            summ = json.load(f)
        for name, k in summ["kernels"].items():
            if name.startswith(kernel):
                return float(k["hbm_read_bytes_per_launch"]) + float(k["hbm_write_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(args, wl, pats, rules, text, offs, lex, gpu_value, unit):
    """The oracle (CPU restatement of the reference path) timed on a bounded sample of rank 0's shard.
    The reference's own lexer stage is Intel Hyperscan, which is not available here: the L1 number is
    the scalar NFA restatement ("port"), not Hyperscan."""
    import oracle
    from struspattern_amd import synth
    # a one-GPU box grants about 16 host cores to the job whatever os.cpu_count() says
    ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if wl == "l2":
        o = oracle.L2Matcher()
        synth.apply_rules(o, rules)
        nd = max(1, min(len(offs) - 1, int(args.cpu_seconds * 200000 / max(1, args.docsize))))
        sub = synth.lexems5(lex[:int(offs[nd])])
        t0 = time.perf_counter()
        r = o.run(sub, offs[:nd + 1], nthreads=1)
        dt = time.perf_counter() - t0
        return {"value": len(r.results) / dt, "unit": "matches/s", "cores": 1, "kind": "port", "events_per_s": int(offs[nd]) / dt,
                "sample": "first %d documents of rank 0's shard (%d events), oracle/l2_oracle.cpp, 1 thread, %.1f s" % (nd, int(offs[nd]), dt)}
    ol = oracle.L1Lexer()
    synth.apply_lexer_patterns(ol, pats)
    # calibrate on one document, then size the sample for about cpu_seconds of wall time on all cores
    t0 = time.perf_counter()
    ol.matchDocs(text[:int(offs[1])], offs[:2], nthreads=1)
    t_doc = max(1e-3, time.perf_counter() - t0)
    nd = max(1, min(len(offs) - 1, int(args.cpu_seconds * ncores / t_doc / 1.5)))
    sub_text = text[:int(offs[nd])]
    t0 = time.perf_counter()
    lexems, loffs = ol.matchDocs(sub_text, offs[:nd + 1], nthreads=min(ncores, nd))
    t1 = time.perf_counter()
    out = {"value": len(sub_text) / (t1 - t0) / 1e9, "unit": "GB/s", "cores": min(ncores, nd), "kind": "port",
           "sample": "first %d documents of rank 0's shard (%d bytes), oracle lexer (scalar NFA restatement, NOT Hyperscan), %d threads, %.1f s" % (
               nd, len(sub_text), min(ncores, nd), t1 - t0)}
    if wl == "pipeline":
        om = oracle.L2Matcher()
        synth.apply_rules(om, rules)
        r = om.run(synth.lexems5(lexems), loffs, nthreads=min(ncores, nd))
        t2 = time.perf_counter()
        out["value"] = len(sub_text) / (t2 - t0) / 1e9
        out["matches_per_s"] = len(r.results) / (t2 - t0)
        out["sample"] += " + oracle automaton %.1f s" % (t2 - t1)
    return out


if __name__ == "__main__":
    main()
