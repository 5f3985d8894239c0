"""bench.py prints ONE JSON line with the fields the driver reads (small assembly, a few steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--contigs", "3000", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["metric"] == "contig-pairs/sec" and r["unit"] == "pairs/s" and r["n_gpus"] == 1
    assert r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - r["config"]["pairs"] / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    # the committed counters must have been measured on THIS build of the library
    assert rf["counters_stale"] is False, rf["counters_stale_what"]
    assert rf["lib_version"].startswith("phyloligo_amd") and " src " in rf["lib_version"]
    rag = r["config"]["ragged_assembly"]
    assert "error" not in rag, rag
    # ragged totals: the general kernels alone (ids 1 / 2; the equal-total kernels would own no tile and are not launched)
    assert rag["metrics"]["JSD"]["kernel_id"] == 1 and rag["metrics"]["BC"]["kernel_id"] == 2 and rag["metrics"]["Eucl"]["kernel_id"] == 4
    assert rag["metrics"]["JSD"]["roofline"]["frac"] > 0 and r["value_ragged_assembly"] == rag["metrics"]["JSD"]["pairs_per_s"]
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert r["value"] > 100 * cb["value"]
    assert cb["cores"] <= cb["host_cpus_visible"] and cb["c1_full"]["metric_calls"] == 1000000
    assert 0 < cb["c1_full"]["same_job_on_the_gpu"]["whole_s"] < cb["c1_full"]["distances_s"]       # BASELINE config 1 on both sides
    assert r["config"]["env_knobs"] == {k: v for k, v in os.environ.items() if k.startswith("PO_")}
    gen = r["config"]["jsd_general_kernel_only"]
    assert gen["roofline"]["frac"] > 0 and gen["pairs_per_s"] < r["value"] * 1.5
    assert r["scaling"] == "weak" and "seed 50001" in r["config"]["workload"]
    # stage 1, H2D, D2H and the container write as separate lines, never inside `value`
    pl = r["config"]["path_lines"]
    assert "error" not in pl, pl
    for key in ("h2d_ms", "stage1_ms", "matrix_ms", "d2h_ms", "container_write_ms", "container_path_ms", "e2e_cli_wall_s"):
        assert pl[key] is not None and pl[key] >= 0 or key == "container_write_ms", (key, pl)
    assert pl["e2e_cli_rc"] == 0 and pl["e2e_cli_container_bytes"] == 3000 * 3000 * 4 == pl["container_bytes"] == pl["d2h_bytes"]
    assert abs(pl["matrix_ms"] - r["ms_per_step"]) < 1e-9
    assert pl["mat_text_bytes"] > 6000 * 6000 * 24 and pl["mat_text_gb_per_s"] > 0.5       # the text .mat writer (reference default output)


def test_bench_multi_rank_path_rehearsal():
    """The N > 1 code path (process group, one all-gather of the counts, tournament work lists, per-rank kernel times)
    with two ranks sharing this one GPU over gloo: numbers are meaningless, the record's shape is what is checked.
    The driver's real run is one rank per GPU over RCCL and defaults to BASELINE config 4 (200 000 contigs)."""
    env = dict(os.environ, PO_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--contigs", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and "seed 200001" in r["config"]["workload"]
    mg = r["config"]["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["kernel_ms_per_rank"]) == 2 and mg["allgather_ms"] > 0
    assert mg["kernel_ms_min"] <= mg["kernel_ms_max"] and mg["allgather_bytes"] == 4096 * 256 * 4 + 4096 * 8
    assert r["config"]["env_knobs"].get("PO_BENCH_REHEARSAL") == "1"
    assert abs(r["value"] - r["config"]["pairs"] / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    assert mg["complete_rows_ms"] > 0 and "complete_rows_error" not in mg      # the optional exchange, behind the record


def test_bench_record_survives_an_exchange_that_does_not_complete():
    """The row-completing exchange of an N > 1 run is optional and has never met a second GPU: it runs last, behind a watchdog.
    With a timeout it cannot meet (1 ms) rank 0 prints the record - `value` and all - with the failure named, and the ranks leave."""
    env = dict(os.environ, PO_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--contigs", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--complete-rows-timeout", "0.001"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    mg = r["config"]["multi_gpu"]
    # (rank 0's own watchdog, or its collective failing because the other rank's watchdog was first)
    assert r["n_gpus"] == 2 and r["value"] > 0 and mg["complete_rows_ms"] is None and mg["complete_rows_error"]


def test_bench_launches_its_own_ranks():
    """plain `python bench.py --gpus 2 ...` - no torchrun on the command line, which is how the driver
    starts the N = 1 run - becomes a launcher (it has not touched the GPU), starts its ranks as a fresh child process and
    relays rank 0's single JSON line and the child's return code.  Rehearsal: both ranks share this one GPU over gloo."""
    env = dict(os.environ, PO_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--contigs", "4096", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["multi_gpu"]["ranks_seen"] == 2 and r["steps"] == 2
    per = r["roofline_per_rank"]
    assert [x["rank"] for x in per] == [0, 1]
    for x in per:
        assert x["bound"] == "hbm" and x["achieved"] > 0 and abs(x["frac"] - x["achieved"] / x["peak"]) < 1e-12
        assert x["pairs"] > 0 and x["kernel_ms"] > 0
    assert abs(sum(x["pairs"] for x in per) - (4096 * 4097 / 2.0)) < 1                  # every pair (and the diagonal) once
    # a failing child is reported through the return code, not swallowed
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--contigs", "4096", "--steps", "1",
                          "--warmup", "0", "--metric", "nope"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert bad.returncode != 0


def test_bench_record_of_the_default_run_has_the_north_star_size():
    """the default run (N = 1, 50 000 contigs) also reports BASELINE config 4's assembly on one GPU with a float32 matrix; read from the
    committed record of this round (running it here would repeat 160 GB of work the bench test above does not need)"""
    path = os.path.join(ROOT, "profiles", "r04_bench_jsd_n50000.json")
    r = json.load(open(path))
    c4 = r["config"]["c4_size_one_gpu_float32"]
    assert "error" not in c4 and "skipped" not in c4, c4
    assert c4["pairs"] == 200000 * 199999 / 2.0
    assert c4["JSD"]["kernel_id"] == 6 and c4["Eucl_int8"]["kernel_id"] == 4 and c4["Eucl_f64_mfma"]["kernel_id"] == 3
    assert c4["Eucl_f64_mfma"]["roofline"]["bound"] == "mfma-f64" and c4["Eucl_f64_mfma"]["roofline"]["frac"] > 0.5


def test_bench_default_workloads_are_the_baseline_configs():
    """bench.py --gpus 1 = BASELINE config 2, --gpus N > 1 = BASELINE config 4 (read from the source: running 200 000
    contigs needs more than one GPU)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'n, cfg_name, seed = 50000, "BASELINE config 2", synthetic.SEEDS["C2"]' in src
    assert 'n, cfg_name, seed = 200000, "BASELINE config 4", synthetic.SEEDS["C4"]' in src
