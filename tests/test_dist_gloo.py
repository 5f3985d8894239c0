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


def _exchange_worker(rank, world, port, n, chunk_bytes, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = RowBlockPlan(n, world, align=16)
        rng = np.random.default_rng(5)
        m = rng.random((n, n))
        full = torch.from_numpy(m + m.T)                      # the symmetric matrix the kernels would produce
        lo, hi = plan.rows(rank)
        slab = torch.full((hi - lo, n), float("nan"), dtype=torch.float64)
        mirrors = []
        for (r0, r1), (c0, c1), kind, peer in plan.work(rank):   # stand-in for plan.compute()
            slab[r0 - lo:r1 - lo, c0:c1] = full[r0:r1, c0:c1]
            mirrors.append(None if kind == "diag" else full[r0:r1, c0:c1].T.contiguous())
        plan.complete_rows(rank, slab, mirrors, dist, chunk_bytes=chunk_bytes)
        ret[rank] = bool(torch.equal(slab, full[lo:hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bytes", [(2, 100, 256 << 20), (3, 100, 256 << 20), (4, 131, 256 << 20),
                                                 (2, 100, 1024), (4, 131, 700)])     # small chunks: several steps per message
def test_complete_rows_exchange(world, n, chunk_bytes):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), n, chunk_bytes, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_tournament_covers_every_pair_once():
    for n in (300, 1024, 1000):
        for world in (1, 2, 3, 4, 5, 8):
            plan = RowBlockPlan(n, world)
            cover = np.zeros((n, n), dtype=np.int32)
            for g in range(world):
                for (r0, r1), (c0, c1), kind, peer in plan.work(g):
                    if kind == "diag":
                        cover[r0:r1, c0:c1] += 1
                    else:
                        assert peer is not None and peer != g
                        cover[r0:r1, c0:c1] += 1
                        cover[c0:c1, r0:r1] += 1
            assert (cover == 1).all(), (n, world)
    plan = RowBlockPlan(141312, 8)
    ev = [plan.pair_evaluations(g) for g in range(8)]
    assert max(ev) / (sum(ev) / 8) < 1.001 and sum(ev) == 141312 * 141313 // 2


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
