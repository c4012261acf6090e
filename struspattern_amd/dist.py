"""Multi-GPU plumbing of the path: documents are independent, so they shard across ranks with no
data-path exchange; the only collective is one all-reduce(sum) of the counters at the end
(RCCL over xGMI on the GPUs, gloo in the CPU tests).  SURVEY.md 8(e)."""
import numpy as np


def shard_documents(doc_offsets, rank, world):
    """Contiguous, byte-balanced range of documents for `rank`: returns (first_doc, last_doc_exclusive).
    The ranges of all ranks partition [0, ndocs) in order, so the global output order is
    reconstructible by concatenating the ranks' outputs."""
    doc_offsets = np.asarray(doc_offsets, dtype=np.uint64)
    ndocs = len(doc_offsets) - 1
    total = int(doc_offsets[-1])
    lo_target = total * rank // world
    hi_target = total * (rank + 1) // world
    first = int(np.searchsorted(doc_offsets, lo_target, side="left"))
    last = int(np.searchsorted(doc_offsets, hi_target, side="left"))
    if rank == 0:
        first = 0
    if rank == world - 1:
        last = ndocs
    return min(first, ndocs), min(max(last, first), ndocs)


def reduce_counters(counters, device=None):
    """all-reduce(sum) of a dict of integer counters over the default process group (int64: exact);
    identity when torch.distributed is not initialised (single GPU)."""
    import torch
    import torch.distributed as dist
    keys = sorted(counters)
    if not (dist.is_available() and dist.is_initialized()):
        return {k: int(counters[k]) for k in keys}
    t = torch.tensor([int(counters[k]) for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def reduce_step(counters, seconds, device=None):
    """What bench.py does at the end of the timed region: the job's counters are the sums over the
    ranks (the path's only collective), the job's time is the slowest rank's.
    Returns (summed counters, max seconds)."""
    import torch
    import torch.distributed as dist
    tot = reduce_counters(counters, device)
    if not (dist.is_available() and dist.is_initialized()):
        return tot, float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return tot, float(t[0])
