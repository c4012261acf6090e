#!/usr/bin/env python3
"""bench.py -- one JSON line per run (contract: README of the driver).

A "step" is one pass of the hot path over one batch of synthetic documents that is already
resident in HBM.  One process per GPU (torch.distributed over RCCL when launched with torchrun);
documents shard across ranks with no data-path collective (weak scaling: fixed work per GPU),
the only collective is the final reduce of the counters.

Workloads:
  l2        rule automaton only: BASELINE.json configs[2] (10k two-term rules, 10k docs x 1000 tokens)
  pipeline  lexer + rule automaton (BASELINE.json configs[4] per-GPU shard), once the lexer is built
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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SPA_BENCH_WORKLOAD", "l2"))
    ap.add_argument("--rules", type=int, default=10000)
    ap.add_argument("--docs", type=int, default=10000)
    ap.add_argument("--docsize", type=int, default=1000)
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--op", default="", help="fix the rule operator (default: the 5-way Zipf mix)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-docs", type=int, default=1500, help="documents in the bounded CPU-baseline sample")
    return ap.parse_args()


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

    # ---- build the compiled tables (identical on every rank) and this rank's document shard
    rules = synth.random_rules(args.rules, args.features, seed=2, op=(args.op or None))
    inst = spa.PatternMatcherInstance()
    synth.apply_rules(inst, rules)
    lex, offs = synth.random_documents(args.docs, args.docsize, args.features, seed=1000 + rank)
    nlex = len(lex)
    ctx = inst.createContext(local_rank)
    d_lex = torch.from_numpy(lex.view(np.int32)).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.matchDocsDevice(d_lex.data_ptr(), d_offs.data_ptr(), args.docs, nlex, stream)

    # first pass sizes the output buffers (results beyond the capacity are only counted)
    for _ in range(8):
        step()
        c = ctx.batchCounters()
        if c["failed_docs"] == 0:
            break
        st = ctx.batchStatus(args.docs)
        codes = sorted(set(int(x) for x in st[st != 0]))
        if 9 in codes:      # SP_DOC_ERR_OUTPUT
            ctx.reserveOutput(int(c["results"] * 1.2) + 1024, int(c["items"] * 1.2) + 1024)
        if 2 in codes:      # SP_DOC_ERR_ARENA
            ctx.growArena()
        if not (set(codes) <= {2, 9}):
            raise SystemExit("bench: documents failed with status codes %s" % codes)
    if c["failed_docs"]:
        raise SystemExit("bench: %d documents failed" % c["failed_docs"])
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)  # read after the sync, the events are recorded on the stream
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last_ms = ctx.lastKernelMs()
    counters = ctx.batchCounters()

    # per-launch kernel durations (HIP events on the launch stream): rerun K launches, one event pair each
    kms = []
    for _ in range(args.steps):
        step()
        kms.append(ctx.lastKernelMs())
    kernel_ms = float(np.mean(kms))

    tot = torch.tensor([dt, float(counters["events"]), float(counters["results"])], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    events = float(tot[1])
    results = float(tot[2])

    if rank == 0:
        per_launch_bytes = 16.0 * counters["events"] + 36.0 * counters["results"]   # B_L2
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "matches/s, 10k-rule automaton over a pre-tokenized event stream (lexer stage pending: GB/s of text not yet reported)",
            "value": results * args.steps / dt,
            "unit": "matches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "randomTokenPatternMatch: %d rules (%s), %d docs x %d tokens per GPU, %d features" % (
                args.rules, args.op or "5-op Zipf mix", args.docs, args.docsize, args.features),
                "events_per_step_per_gpu": int(counters["events"]), "matches_per_step_per_gpu": int(counters["results"])},
            "events_per_s": events * args.steps / dt,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "spa_l2_match_kernel", "kernel_ms": kernel_ms},
        }
        if not args.no_cpu_baseline and world >= 1:
            out["cpu_baseline"] = cpu_baseline(args, rules, lex, offs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, rules, lex, offs):
    """The oracle (CPU restatement of the reference automaton) timed on a bounded sample, 1 thread."""
    import oracle
    from struspattern_amd import synth
    o = oracle.L2Matcher()
    synth.apply_rules(o, rules)
    nd = min(args.cpu_docs, len(offs) - 1)
    sub = synth.lexems5(lex[:int(offs[nd])])
    t0 = time.perf_counter()
    r = o.run(sub, offs[:nd + 1], nthreads=1)
    dt = time.perf_counter() - t0
    return {"value": len(r.results) / dt, "unit": "matches/s", "cores": 1, "kind": "port",
            "events_per_s": int(offs[nd]) / dt,
            "sample": "first %d documents of rank 0's shard (%d events), oracle/l2_oracle.cpp, 1 thread, %.1f s" % (nd, int(offs[nd]), dt)}


if __name__ == "__main__":
    main()
