"""The `nccl` (= RCCL) backend inside the GPU test run.  A one-GPU box can hold one rank, so what is exercised is the
API surface the multi-GPU path depends on (process group on a device, all_gather_into_tensor of int32 / int64 device
tensors, batched isend / irecv of float64 chunks, all_reduce, barrier) -- not scaling.  Reference shard shape:
gen_even_slices row blocks, /root/reference/phylopackage/bin/phyloligo.py:424."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_calls_of_the_multi_gpu_path_one_rank():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=_env(29541))
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["allgather_new_buffers"] and r["allgather_equal"]
    assert r["p2p_equal"] and r["complete_rows_noop"]
    assert r["all_reduce"] == [2.5, 1, 2.5]


def test_bench_over_rccl_one_rank():
    """bench.py's distributed path (PO_BENCH_FORCE_DIST=1) on the nccl backend with one rank: process group, the
    all-gather of the counts through RCCL, barriers around the timed region, the all-reduces of the record."""
    env = _env(29543)
    env["PO_BENCH_FORCE_DIST"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--contigs", "4096", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    mg = r["config"]["multi_gpu"]
    assert mg["backend"] == "nccl (RCCL)" and mg["ranks_seen"] == 1
    assert mg["allgather_ms"] is not None and mg["allgather_ms"] > 0
    assert mg["allgather_bytes"] == 4096 * 256 * 4 + 4096 * 8
    assert r["n_gpus"] == 1 and r["config"]["env_knobs"].get("PO_BENCH_FORCE_DIST") == "1"
