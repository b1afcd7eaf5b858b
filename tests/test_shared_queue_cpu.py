"""The host side of the scenario queue shared between ranks (sqpsolver.jl_amd/shard.py: rebalance_plan,
run_shared_queue) on the CPU: the plan, and the whole protocol over a gloo process group of two ranks with simulated
device queues -- every scenario is solved exactly once, by one rank, and the rank that runs dry is fed."""
import multiprocessing as mp
import os

import numpy as np

from sqpsolver_jl_amd.shard import rebalance_plan, run_shared_queue


def test_rebalance_plan_feeds_the_hungry_rank():
    assert rebalance_plan([0, 10], [0, 4], [4, 4]) == [(1, 0, 4)]
    assert rebalance_plan([0, 10], [4, 4], [4, 4]) == []            # no idle slot: nothing to hand over
    assert rebalance_plan([3, 3], [4, 4], [4, 4]) == []             # nobody hungry
    assert rebalance_plan([0, 1], [0, 4], [4, 4]) == []             # a single id is not worth a transfer
    assert rebalance_plan([0, 0, 9], [0, 2, 4], [4, 4, 4]) == [(2, 0, 4), (2, 1, 2)]


class FakeQueue:
    """stands in for a context: each scenario needs `cost[id]` outer iterations; `slots` slots"""
    def __init__(self, ids, cost, slots):
        self.q = list(ids); self.cost = cost; self.slots = slots; self.running = {}; self.solved = []

    def run_some(self, k):
        for _ in range(k):
            while len(self.running) < self.slots and self.q:
                s = self.q.pop(0); self.running[s] = self.cost[s]
            for s in list(self.running):
                self.running[s] -= 1
                if self.running[s] == 0:
                    del self.running[s]; self.solved.append(s)
        return len(self.q), len(self.running)

    def release(self, n):
        n = min(n, len(self.q)); out = self.q[len(self.q) - n:]; del self.q[len(self.q) - n:]
        return out

    def append(self, ids):
        self.q.extend(ids)


def _rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    M = 40
    cost = np.random.default_rng(0).integers(3, 30, size=M).tolist()
    ids = list(range(0, 34)) if rank == 0 else list(range(34, 40))      # an uneven split: rank 1 runs dry early
    q = FakeQueue(ids, cost, slots=4)
    rounds = run_shared_queue(q, rank, world, 4, chunk=5)
    out.put((rank, sorted(q.solved), rounds))
    dist.destroy_process_group()


def test_shared_queue_protocol_over_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29611 + os.getpid() % 200
    ps = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(out.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res[0][1], res[1][1]
    assert sorted(s0 + s1) == list(range(40))                 # every scenario exactly once
    assert len(s1) > 6                                        # rank 1 was fed from rank 0's queue
    assert res[0][2] == res[1][2]                             # both ranks leave the loop in the same round
