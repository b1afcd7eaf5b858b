"""CPU tests of the N>1 path: contiguous sharding of independent instances and the status
all-gather, with a world_size-2 `gloo` process group (the GPU path uses the same code over nccl/RCCL)."""
import os
import socket

import numpy as np
import pytest

from sqpsolver_jl_amd.shard import shard_range, gather_status


def test_shard_range_partitions_exactly():
    for total, world in ((512, 8), (64, 1), (10, 4), (3, 8), (0, 2)):
        seen = []
        for r in range(world):
            lo, hi = shard_range(total, world, r)
            assert 0 <= lo <= hi <= total
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    assert shard_range(512, 8, 3) == (192, 256)               # 64 scenarios per GPU (BASELINE config 4)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_gather_without_process_group_is_identity():
    ret = np.array([0, -1, 6], dtype=np.int32); it = np.array([5, 3000, 17], dtype=np.int32)
    done = np.array([1, 1, 0], dtype=np.int32)
    a, b, c = gather_status(ret, it, done, 3)
    assert a.tolist() == [0, -1, 6] and b.tolist() == [5, 3000, 17] and c.tolist() == [1, 1, 0]


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, world, rank)
        ids = np.arange(lo, hi)
        ret = (ids % 7 - 3).astype(np.int32)                  # fake return codes keyed by global id
        it = (100 + ids).astype(np.int32)
        done = (ids % 2).astype(np.int32)
        g = gather_status(ret, it, done, total)
        q.put((rank, [a.tolist() for a in g]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_status_gather_world2_gloo(total):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = np.arange(total)
    want = [(ids % 7 - 3).tolist(), (100 + ids).tolist(), (ids % 2).tolist()]
    for _, got in outs:                                       # every rank ends with the full table
        assert got == want
