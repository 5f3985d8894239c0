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
        # the plan's own buffers: rows on 128-byte boundaries, i.e. strided views whenever a width is not a multiple of 32 entries
        # (n = 100, 131 and the 16-aligned blocks here) - the exchange has to cope with them (round 5)
        slab, mirrors = plan.allocate(rank, "cpu", torch.float64)
        slab.fill_(float("nan"))
        assert slab.shape == (hi - lo, n) and (n % 32 == 0 or hi - lo < 2 or not slab.is_contiguous())
        for ((r0, r1), (c0, c1), kind, peer), mir in zip(plan.work(rank), mirrors):   # stand-in for plan.compute()
            slab[r0 - lo:r1 - lo, c0:c1] = full[r0:r1, c0:c1]
            if kind != "diag":
                mir.copy_(full[r0:r1, c0:c1].T)
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


def test_launcher_starts_ranks_relays_output_and_status(tmp_path):
    """phyloligo_amd.launch.spawn_ranks (what `bench.py --gpus N` and `python -m phyloligo_amd --gpus N` become when started
    plainly): ranks run as a fresh child process group, write to the launcher's stdout, the return code comes back, a hung
    job is killed as a group on timeout, and the launcher itself never imports torch.  No GPU involved: the ranks are a
    tiny script joining a gloo group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rank_script = tmp_path / "rank.py"
    rank_script.write_text(
        "import os, sys, time\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.ones(1); dist.all_reduce(t)\n"
        "if sys.argv[1] == 'hang':\n"
        "    time.sleep(600)\n"
        "if dist.get_rank() == 0:\n"
        "    print('RANKS %d ARG %s' % (int(t.item()), sys.argv[1]), flush=True)\n"
        "dist.destroy_process_group()\n"
        "sys.exit(3 if sys.argv[1] == 'fail' else 0)\n")
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from phyloligo_amd import launch\n"
              "assert launch.needs_launcher(2) and not launch.needs_launcher(1)\n"
              "rc = launch.spawn_ranks(2, [%r], [sys.argv[1]], timeout_s=float(sys.argv[2]))\n"
              "assert 'torch' not in sys.modules\n"
              "sys.exit(rc)\n") % (root, str(rank_script))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    ok = subprocess.run([sys.executable, "-c", driver, "fine", "120"], capture_output=True, text=True, timeout=300, env=env)
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert ok.stdout.count("RANKS 2 ARG fine") == 1
    bad = subprocess.run([sys.executable, "-c", driver, "fail", "120"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0
    t0 = __import__("time").time()
    hung = subprocess.run([sys.executable, "-c", driver, "hang", "20"], capture_output=True, text=True, timeout=300, env=env)
    assert hung.returncode == 124 and "killing the process group" in hung.stderr
    assert __import__("time").time() - t0 < 120
    # inside a rank (WORLD_SIZE set) nobody launches again
    env2 = dict(env, WORLD_SIZE="2")
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nfrom phyloligo_amd import launch\n"
                          "sys.exit(1 if launch.needs_launcher(8) else 0)" % root], env=env2, timeout=60)
    assert chk.returncode == 0


def test_first_contact_failure_is_one_json_line_and_a_status(tmp_path):
    """no process group of more than one GPU has ever run this code, so the first 8-GPU run must fail
    LEGIBLY if it fails - rank 0 prints one JSON line with the stage, the error and the environment (HSA_ENABLE_IPC_MODE_LEGACY
    named even when unset, RCCL / torch versions), every rank exits with status 3 and the launcher hands a non-zero status on.
    Driven on the CPU with a backend that does not exist, two ranks under launch.spawn_ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rank_script = tmp_path / "rank.py"
    rank_script.write_text(
        "import sys; sys.path.insert(0, %r)\n"
        "import torch.distributed as dist\n"
        "from phyloligo_amd.dist import first_contact\n"
        "first_contact('init_process_group', dist.init_process_group, sys.argv[1])\n"
        "ok = first_contact('all_gather_profiles', lambda: 7)\n"
        "print('CONTACT', ok, flush=True)\n"
        "dist.destroy_process_group()\n" % root)
    driver = ("import sys; sys.path.insert(0, %r)\n"
              "from phyloligo_amd import launch\n"
              "sys.exit(launch.spawn_ranks(2, [%r], [sys.argv[1]], timeout_s=120.0))\n") % (root, str(rank_script))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "HSA_ENABLE_IPC_MODE_LEGACY")}
    bad = subprocess.run([sys.executable, "-c", driver, "no_such_backend"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0
    lines = [json.loads(l) for l in bad.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, bad.stdout[-2000:]                       # rank 0 alone writes to stdout
    rec = lines[0]
    assert rec["stage"] == "init_process_group" and rec["rank"] == 0 and rec["world_size"] == 2 and "no_such_backend" in rec["error"]
    assert rec["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # what launch.spawn_ranks put there (nobody had set it)
    assert rec["env"]["WORLD_SIZE"] == "2" and "MASTER_PORT" in rec["env"] and "torch" in rec and "rccl" in rec
    # (the other rank reports on stderr - if torch.distributed.run has not taken it down first)
    ok = subprocess.run([sys.executable, "-c", driver, "gloo"], capture_output=True, text=True, timeout=300, env=env)
    assert ok.returncode == 0 and ok.stdout.count("CONTACT 7") == 2, ok.stderr[-2000:]
