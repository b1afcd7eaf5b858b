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


# ---- a scenario queue shared between ranks --------------------------------------------------
def rebalance_plan(unstarted, active, slots):
    """Which unstarted scenario ids move where, from the gathered queue state of all ranks (every rank computes the same
    plan from the same numbers).  unstarted[r]: ids of rank r's queue nobody has drawn yet; active[r]: slots of rank r
    with a run in progress; slots[r]: slots of rank r.  A rank is HUNGRY when its queue is empty and it has idle slots;
    it is served from the rank with the most unstarted ids, which keeps at least as many as it gives away.  Returns a
    list of (src, dst, k), in the order the transfers are carried out."""
    u = [int(v) for v in unstarted]
    plan = []
    hungry = sorted((r for r in range(len(u)) if u[r] == 0 and active[r] < slots[r]), key=lambda r: (active[r], r))
    for dst in hungry:
        src = max(range(len(u)), key=lambda r: (u[r], -r))
        if u[src] < 2:
            break
        k = min(u[src] // 2, int(slots[dst]) - int(active[dst]))
        if k <= 0:
            continue
        plan.append((src, dst, k))
        u[src] -= k
        u[dst] += k
    return plan


def run_shared_queue(queue, rank, world, slots, chunk=5, exchange=None, max_rounds=100000):
    """Drive one rank's part of a scenario queue shared between `world` ranks until every scenario of the job has been
    solved.  `queue` is this rank's context (host.Context after stream_begin / stream_set of ALL scenarios and
    stream_assign of this rank's ids) or anything with run_some(k) -> (unstarted, active), release(n) -> ids,
    append(ids).  Per round: every rank runs `chunk` outer iterations per slot, the ranks exchange two integers each,
    and unstarted ids move from the fullest queue to ranks whose queue ran dry (rebalance_plan); the ids themselves
    travel in a second small exchange.  `exchange(list_of_ints) -> list over ranks of lists` is the host channel
    (default: torch.distributed all_gather_object).  Returns the number of rounds."""
    if exchange is None:
        import torch.distributed as dist

        def exchange(obj):
            if world == 1:
                return [obj]
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
    run_some = getattr(queue, "stream_run_some", None) or queue.run_some
    release = getattr(queue, "stream_release", None) or queue.release
    append = getattr(queue, "stream_append", None) or queue.append
    for rnd in range(1, max_rounds + 1):
        u, a = run_some(chunk)
        state = exchange([int(u), int(a), int(slots)])
        U, A, S = [s[0] for s in state], [s[1] for s in state], [s[2] for s in state]
        if sum(U) == 0 and sum(A) == 0:
            return rnd
        plan = rebalance_plan(U, A, S)
        if not plan:
            continue
        given = [[int(v) for v in release(k)] if src == rank else [] for (src, dst, k) in plan]
        moved = exchange(given)                  # moved[r][t]: the ids rank r released for transfer t
        for t, (src, dst, k) in enumerate(plan):
            if dst == rank and moved[src][t]:
                append(moved[src][t])
    raise RuntimeError("run_shared_queue: no termination")
