"""Batch sharding of independent NLP instances over ranks and the convergence-status gather.

The reference is a single process (no distributed code anywhere in /root/reference); the scaling
axis of the hot path is a batch of independent instances (ACOPF contingency scenarios, SURVEY.md
section 8e).  Instances are cut into contiguous blocks, one per rank (one rank per GPU); no iterate
ever crosses ranks.  The only exchange is an all-gather of (return code, iteration count, done flag)
per instance -- `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of instance ids owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_status(ret: np.ndarray, iters: np.ndarray, done: np.ndarray, total: int, device=None):
    """All-gather per-instance (ret, iter, done) triples; returns three int32 arrays of length
    `total` ordered by global instance id.  Works without an initialised process group (world 1)."""
    import torch
    import torch.distributed as dist

    local = np.stack([ret, iters, done], axis=1).astype(np.int32)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        full = local
    else:
        world = dist.get_world_size()
        sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
        cap = max(sizes)
        buf = torch.zeros((cap, 3), dtype=torch.int32, device=device)
        buf[: local.shape[0]] = torch.from_numpy(local).to(buf.device)
        out = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        full = np.concatenate([o[: sizes[r]].cpu().numpy() for r, o in enumerate(out)], axis=0)
    assert full.shape[0] == total
    return full[:, 0].copy(), full[:, 1].copy(), full[:, 2].copy()
