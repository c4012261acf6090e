"""world_size-2 test of the multi-GPU plumbing on CPU (gloo): byte-balanced contiguous document
shards partition the corpus, and the counter reduce sums the per-rank counters."""
import os
import socket

import numpy as np

from struspattern_amd import dist as spdist


def test_shards_partition_the_corpus():
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        lens = rng.integers(0, 70000, size=257)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        prev = 0
        sizes = []
        for r in range(world):
            a, b = spdist.shard_documents(offs, r, world)
            assert a == prev and b >= a
            prev = b
            sizes.append(int(offs[b] - offs[a]))
        assert prev == len(lens)
        assert max(sizes) - min(sizes) <= 2 * 70000


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs = np.arange(0, 101, dtype=np.uint64) * 10
        a, b = spdist.shard_documents(offs, rank, world)
        local = {"bytes": int(offs[b] - offs[a]), "lexems": (b - a) * 3, "matches": rank + 1}
        tot = spdist.reduce_counters(local)
        # bench.py's end of the timed region: counters summed (exact above 2^53 too), time maxed
        big, dt = spdist.reduce_step({"bytes": (1 << 60) + rank, "results": 7 * (rank + 1)}, 1.5 + rank)
        assert big == {"bytes": (1 << 61) + 1, "results": 21} and dt == 2.5
        out.put((rank, a, b, tot))
    finally:
        dist.destroy_process_group()


def test_counter_reduce_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a0, b0, t0), (r1, a1, b1, t1) = res
    assert (a0, b1) == (0, 100) and b0 == a1
    assert t0 == t1 == {"bytes": 1000, "lexems": 300, "matches": 3}
