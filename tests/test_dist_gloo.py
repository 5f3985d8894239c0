"""The N>1 path on CPU: two processes over gloo exercise the row-block plan and the single
all-gather of the exact count matrix (the only exchange step of the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phyloligo_amd.dist import RowBlockPlan


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, dim, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = RowBlockPlan(n, world)
        lo, hi = plan.rows(rank)
        rng = np.random.default_rng(123)
        full_c = rng.integers(0, 50, size=(n, dim)).astype(np.int32)
        full_t = full_c.sum(axis=1).astype(np.int64)
        counts, totals = plan.all_gather_profiles(torch.from_numpy(full_c[lo:hi].copy()),
                                                  torch.from_numpy(full_t[lo:hi].copy()), dist)
        ok = np.array_equal(counts.numpy(), full_c) and np.array_equal(totals.numpy(), full_t)
        ret[rank] = (ok, lo, hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [256, 1000, 131])
def test_all_gather_profiles_world2(n):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, 16, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
    spans = sorted((ret[r][1], ret[r][2]) for r in range(world))
    assert spans[0][0] == 0 and spans[-1][1] == n and spans[0][1] == spans[1][0]


def test_row_blocks_partition():
    for n in (1, 127, 128, 129, 50000, 70711, 200000):
        for world in (1, 2, 3, 4, 8):
            plan = RowBlockPlan(n, world)
            covered = 0
            for r in range(world):
                lo, hi = plan.rows(r)
                assert lo <= hi and lo == covered
                assert lo % 128 == 0 or lo == n
                covered = hi
            assert covered == n
