#!/usr/bin/env python3
"""Headline benchmark: contig-pairs/sec of the all-by-all distance matrix (BASELINE.json).

A step = one pass of stage 2 (counts resident in HBM -> every matrix entry of this rank's shard
resident in HBM, float64) over one synthetic assembly.

  --gpus 1   BASELINE config 2: 50 000 contigs x 2 kb (seed 50001), k=4, both strands, -d JSD.
  --gpus N>1 BASELINE config 4: 200 000 contigs x 2 kb (seed 200001), k=4, -d JSD, row-block sharded: one
             process per GPU (torch.distributed, RCCL); every rank profiles its own contigs, the exact count
             matrix is all-gathered ONCE, then the upper triangle of the block grid is dealt out tournament-style
             (phyloligo_amd/dist.py) so that every pair is evaluated once, with no exchange inside the timed
             region.  The matrix (320 GB in float64) does not fit one GPU, which is why N=1 runs config 2; the
             work is the same for N = 2, 4, 8 ("strong" scaling), and pairs/s is comparable across all N.

Beside `value` the JSON line carries: the roofline of the dominant kernel (HIP-event time on the launch stream),
the same for the other single-GPU BASELINE configs (C3 Eucl on both its paths, C5 BC) and for the general JSD
kernel that ragged assemblies get, the CPU baseline (oracle = the reference's joblib path restated) and, for N>1,
the times of the single exchange (all-gather) and of the optional row-completing exchange.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable by a float4 copy)
F64_MFMA_PEAK_TF = 78.6   # v_mfma_f64_16x16x4_f64 dense peak

KERNEL_NAMES = {1: "valu_tile_kernel<JSD>", 2: "valu_tile_kernel<BC>", 3: "gram_tile_kernel (f64 MFMA)",
                4: "gram_i8_quad|half|tile_kernel (exact int8 MFMA)", 5: "kt_tile_kernel",
                6: "jsd_lut_rows_kernel (equal-total record blocks) + valu_tile_kernel<JSD> (rest)",
                7: "bc_sad_tile_kernel (equal-total record blocks) + valu_tile_kernel<BC> (rest)",
                8: "pairdot_tile_kernel<KT> (materialised pair-sign Gram on the matrix cores)",
                9: "pairdot_tile_kernel<BC> (thermometer planes: sum of min on the matrix cores)"}


def _oracle_rows(freq, metric, rows):
    from oracle import phyloligo_oracle as po
    return po.pairwise_rows(freq, metric, rows).shape


def _oracle_profiles(seqs):
    from oracle import phyloligo_oracle as po
    return po.compute_frequencies(seqs, "1111", "both")


def _oracle_profiles_per_window(seqs):
    """the reference's own way: one Python string per window (phyloligo.py:622-631)"""
    from oracle import phyloligo_oracle as po
    t0 = time.perf_counter()
    for s in seqs:
        po.profile_counts_per_window(s, "1111", "both")
    return time.perf_counter() - t0


def usable_cores():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup CPU quota when there is one
    (a GPU box shows all 256 hardware threads of the host to os.cpu_count() but gives a one-GPU job a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = fh.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _timed_rows(freq, metric, row0, budget_s):
    """one worker: rows row0, row0 + 1, ... of the oracle's per-pair loop until budget_s is used up"""
    from oracle import phyloligo_oracle as po
    t0 = time.perf_counter()
    rows = 0
    while True:
        po.pairwise_rows(freq, metric, [(row0 + rows) % freq.shape[0]])
        rows += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or rows >= 64:
            return rows, dt


def cpu_baseline(freq, metric, budget_s=10.0, cores=None):
    """The oracle's per-pair Python path (= the reference's joblib path: one metric call per pair under
    sklearn.pairwise_distances, phyloligo.py:364-392) on every CPU this job may use.
    (1) a time-bounded slab of rows x all columns of THIS workload (each worker evaluates whole rows until budget_s is
    used up); (2) BASELINE config 1 in full (1 000 contigs, Eucl: profiles + 10^6 metric calls), as SURVEY 8d asks."""
    import joblib
    from oracle import phyloligo_oracle as po
    cores = cores or usable_cores()
    n = freq.shape[0]
    with joblib.Parallel(n_jobs=cores) as par:
        par(joblib.delayed(_oracle_rows)(freq[:8], metric, [0]) for _ in range(cores))          # start the workers
        t0 = time.perf_counter()
        done = par(joblib.delayed(_timed_rows)(freq, metric, c * 64, budget_s) for c in range(cores))
        dt = time.perf_counter() - t0
        rows = sum(r for r, _ in done)
        evaluated = rows * n
        out = {"value": evaluated / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
               "host_cpus_visible": os.cpu_count(),
               "sample": "%d rows x %d columns = %d metric calls of oracle.%s (per-pair numpy path of phylodist.py under "
                         "sklearn.pairwise_distances semantics) in %.1f s over %d processes"
                         % (rows, n, evaluated, metric, dt, cores)}
        # ---- BASELINE config 1, whole: 1 000 contigs x 2 kb, k=4 both strands, -d Eucl ----
        seqs = po.synthetic_contigs(1000, 2000, seed=1001)
        per = -(-len(seqs) // cores)
        t0 = time.perf_counter()
        parts = par(joblib.delayed(_oracle_profiles)(seqs[c * per:(c + 1) * per]) for c in range(cores) if seqs[c * per:(c + 1) * per])
        f1 = np.vstack(parts)
        t_prof = time.perf_counter() - t0
        per = -(-1000 // cores)
        t0 = time.perf_counter()
        par(joblib.delayed(_oracle_rows)(f1, "Eucl", list(range(c * per, min(1000, (c + 1) * per)))) for c in range(cores)
            if c * per < 1000)
        t_dist = time.perf_counter() - t0
        # stage 1 as the reference runs it - a Python string per window - on 4 contigs per process, scaled to the 1 000
        sample = 4
        t0 = time.perf_counter()
        per_core = par(joblib.delayed(_oracle_profiles_per_window)(seqs[c * sample:(c + 1) * sample]) for c in range(cores))
        t_pw = (time.perf_counter() - t0) * (1000.0 / (sample * cores))
    out["c1_full"] = {"workload": "BASELINE config 1: 1 000 contigs x 2 kb (seed 1001), k=4 both strands, -d Eucl, whole job",
                      "profiles_s": t_prof, "profiles_kind": "oracle.count_pattern: vectorised numpy (one bincount per record), NOT the "
                      "reference's per-window Python loop - see profiles_per_window_s",
                      "profiles_per_window_s": t_pw, "profiles_per_window_kind": "oracle.profile_counts_per_window (a Python string per "
                      "window, as bin/phyloligo.py:622-631): %d contigs per process timed on %d processes, scaled to 1 000; "
                      "per-core seconds per contig: %.4f" % (sample, cores, sum(per_core) / (sample * cores)),
                      "distances_s": t_dist, "pairs_per_s": 499500.0 / t_dist,
                      "metric_calls": 1000 * 1000, "cores": cores}
    return out


def hbm_roofline(algo_bytes, ms, **extra):
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    d = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    d.update(extra)
    return d


def path_lines(ctx, seq, offsets, counts, totals, n, dim, metric, pattern, dev, stage1_ms, matrix_ms):
    """Every step of the path either side of the timed matrix, each as its own line (reference: main(),
    /root/reference/phylopackage/bin/phyloligo.py:1036-1068 - read the assembly, profile, distances, write).
    h2d: sequence bytes + offsets host -> HBM; d2h: the float32 matrix (the --large memmap container type,
    bin/phyloligo.py:413) HBM -> pageable host memory through po_pairwise's pinned ring; container: compute_distances(...,
    large="memmap") = row blocks computed, copied and pwrite()n into the raw float32 file; e2e: the CLI itself in a child
    process on a FASTA file of this assembly (process start and imports included)."""
    import shutil
    import subprocess
    import tempfile
    import torch
    from phyloligo_amd import api
    from phyloligo_amd import phyloligo as P
    from phyloligo_amd import synthetic
    lines = {"workload": "%d contigs, dim %d, -d %s" % (n, dim, metric), "stage1_ms": stage1_ms, "matrix_ms": matrix_ms,
             "note": "none of these is part of `value` (matrix resident in HBM)"}
    # ---- H2D of the assembly ----
    off64 = offsets.astype(np.int64)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        d_seq = torch.from_numpy(seq).to(dev)
        d_off = torch.from_numpy(off64).to(dev)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    del d_seq, d_off
    lines["h2d_ms"] = best
    lines["h2d_bytes"] = int(seq.nbytes + off64.nbytes)
    # ---- D2H of the float32 matrix through the host-pointer entry point (H2D of the counts + matrix + D2H ring) ----
    c_h, t_h = counts.cpu().numpy().astype(np.uint32), totals.cpu().numpy().astype(np.uint64)
    need = n * n * 4
    import psutil
    if psutil.virtual_memory().available > 3 * need:
        first = None
        for _ in range(2):                                        # the first call also grows the 10 GB device staging buffer
            host = np.empty((n, n), dtype=np.float32)             # a fresh, untouched destination each time (as a caller's would be)
            t0 = time.perf_counter()
            res, st = ctx.pairwise(c_h, t_h, metric, dtype="float32", out=host, want_stats=True)
            wall = (time.perf_counter() - t0) * 1e3
            if first is None:
                first = wall
            # both names: rounds 2 - 4 kept the 10 GB result alive under the name `_` and freed it inside the NEXT timed region -
            # "ingest_and_profiles_ms 490 - 590" was 0.45 s of munmap plus 35 - 60 ms of ingest (cProfile of the call: 31 ms)
            del host, res
        host = None
        lines["host_pointer_first_call_ms"] = first
        lines["d2h_ms"] = wall - st["total_ms"]
        lines["d2h_bytes"] = int(need)
        lines["d2h_gb_per_s"] = need / max(1e-9, (wall - st["total_ms"]) * 1e-3) / 1e9
        lines["host_pointer_call_ms"] = wall
        lines["pairs_per_s_pcie_inclusive"] = n * (n - 1) / 2.0 / (wall * 1e-3)
    else:
        lines["d2h_ms"] = None
        lines["d2h_note"] = "skipped: not enough host memory for a %d-byte result" % need
    ctx.trim()                                                    # the 10 GB staging of the host-pointer form goes back
    # ---- the raw float32 container and the CLI end to end, on a FASTA file of this assembly ----
    tmp_root = os.environ.get("TMPDIR") or tempfile.gettempdir()
    if shutil.disk_usage(tmp_root).free < need + seq.nbytes * 2 + (1 << 30):
        lines["container_write_ms"] = None
        lines["container_note"] = "skipped: not enough free space under %s" % tmp_root
        return lines
    with tempfile.TemporaryDirectory(dir=tmp_root) as tmp:
        fa = os.path.join(tmp, "assembly.fa")
        with open(fa, "wb") as fh:
            fh.write(synthetic.fasta_bytes(seq, offsets))
        lines["fasta_bytes"] = os.path.getsize(fa)
        P.INGEST_PHASES = True
        t0 = time.perf_counter()
        freq, freq_name = P.compute_frequencies("hip", "memmap", fa, pattern, "both", 250, 4, tmp)
        lines["ingest_and_profiles_ms"] = (time.perf_counter() - t0) * 1e3
        lines["ingest_phases_ms"] = {k: round(v, 3) for k, v in (P.LAST_INGEST or {}).items()}
        t0 = time.perf_counter()
        P.compute_frequencies("hip", "memmap", fa, pattern, "both", 250, 4, tmp)
        lines["ingest_and_profiles_second_call_ms"] = (time.perf_counter() - t0) * 1e3
        lines["ingest_phases_second_call_ms"] = {k: round(v, 3) for k, v in (P.LAST_INGEST or {}).items()}
        P.INGEST_PHASES = False
        out = os.path.join(tmp, "matrix.f32")
        t0 = time.perf_counter()
        P.compute_distances("hip", "memmap", freq, None, out, metric, 4, 250, tmp)
        lines["container_path_ms"] = (time.perf_counter() - t0) * 1e3           # compute + D2H + pwrite, overlapped
        lines["container_bytes"] = os.path.getsize(out)
        lines["container_write_ms"] = lines["container_path_ms"] - matrix_ms
        lines["container_gb_per_s"] = lines["container_bytes"] / (lines["container_path_ms"] * 1e-3) / 1e9
        os.remove(out)
        del freq
        ctx.trim()
        # the reference's default output: the text .mat (numpy.savetxt layout) of a 6 000 x 6 000 corner of the matrix, twice
        # (the second write goes over the existing file: no first-touch of new page-cache pages)
        m = np.random.default_rng(7).random((6000, 6000))
        txt = os.path.join(tmp, "corner.mat")
        for key in ("mat_text_first_write_s", "mat_text_write_s"):
            t0 = time.perf_counter()
            api.write_mat_text(txt, m)
            lines[key] = time.perf_counter() - t0
        lines["mat_text_bytes"] = os.path.getsize(txt)
        lines["mat_text_gb_per_s"] = lines["mat_text_bytes"] / lines["mat_text_write_s"] / 1e9
        os.remove(txt)
        del m
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        cmd = [sys.executable, "-m", "phyloligo_amd", "-i", fa, "-p", pattern, "-d", metric, "--method", "joblib", "--large", "memmap",
               "-o", out, "-w", tmp]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        lines["e2e_cli_wall_s"] = time.perf_counter() - t0
        lines["e2e_cli"] = "python -m phyloligo_amd -i assembly.fa -p %s -d %s --method joblib --large memmap -o matrix.f32" % (pattern, metric)
        lines["e2e_cli_rc"] = r.returncode
        if r.returncode == 0:
            lines["e2e_cli_container_bytes"] = os.path.getsize(out)
        else:
            lines["e2e_cli_stderr"] = r.stderr[-400:]
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=0, help="override the number of contigs (default: 50 000 at 1 GPU = "
                                                           "BASELINE config 2, 200 000 at N>1 = config 4)")
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("--metric", default="JSD", choices=["Eucl", "JSD", "KT", "BC", "SC"])
    ap.add_argument("--pattern", default="1111")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 / KT extras in config.other_configs")
    ap.add_argument("--no-ragged", action="store_true", help="skip config.ragged_assembly (a real-assembly-like input of the same "
                                                             "number of contigs through every metric)")
    ap.add_argument("--no-complete-rows", action="store_true", help="N>1: skip timing the optional row-completing exchange")
    ap.add_argument("--no-path-lines", action="store_true", help="skip config.path_lines (H2D, D2H, container write, end-to-end CLI wall)")
    ap.add_argument("--complete-rows-timeout", type=float, default=300.0,
                    help="seconds the optional row-completing exchange of an N > 1 run may take before the record is printed without it")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("PO_BENCH_LAUNCH_TIMEOUT", "3000")),
                    help="--gpus N>1 started plainly: seconds after which the launcher kills its ranks")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no WORLD_SIZE: this process becomes the launcher (it has imported neither torch
    # nor the library, so it has not touched a GPU) and starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    # as a fresh child; rank 0 of the child prints the JSON line straight to our stdout.  The reference starts its own workers
    # the same way (joblib Parallel over gen_even_slices, bin/phyloligo.py:386-390, :424).
    from phyloligo_amd import launch
    if launch.needs_launcher(args.gpus):
        return launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)], sys.argv[1:], timeout_s=args.launch_timeout)

    import torch
    import phyloligo_amd as pa
    from phyloligo_amd import synthetic
    from phyloligo_amd.dist import RowBlockPlan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Environment switches of this harness and of the library are part of the record: PO_BENCH_* change what is run,
    # any other PO_* would be a kernel-variant knob (none is read by the shipped library any more).
    knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith("PO_")}
    # PO_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo collectives -- lets the N>1 code path be run on
    # a one-GPU box (numbers are meaningless then); the real run is one rank per GPU over RCCL.
    rehearsal = os.environ.get("PO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # PO_BENCH_FORCE_DIST=1: take the torch.distributed path even with one rank (exercises the RCCL calls on a
    # one-GPU box: process group, all_gather_into_tensor, barrier, all_reduce)
    if world > 1 or os.environ.get("PO_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        from phyloligo_amd.dist import first_contact
        first_contact("cuda.set_device", torch.cuda.set_device, local_rank)
        # (PO_BENCH_BACKEND: a backend name for the failure drill of tests/test_dist_gloo.py; unset in every real run)
        backend = os.environ.get("PO_BENCH_BACKEND") or ("gloo" if rehearsal else "nccl")
        if backend == "nccl":
            first_contact("init_process_group", dist.init_process_group, "nccl", device_id=torch.device("cuda", local_rank))
        else:
            first_contact("init_process_group", dist.init_process_group, backend)
    else:
        dist = None
    if world != args.gpus:               # started under torch.distributed.run with a different --nproc-per-node
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (plain `python bench.py --gpus N` starts its own ranks)" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    cdev = "cpu" if rehearsal else dev            # where small collectives live (gloo in rehearsal)

    if args.contigs:
        n, cfg_name, seed = args.contigs, "custom", synthetic.SEEDS["C2" if world == 1 else "C4"]
    elif world == 1:
        n, cfg_name, seed = 50000, "BASELINE config 2", synthetic.SEEDS["C2"]
    else:
        n, cfg_name, seed = 200000, "BASELINE config 4", synthetic.SEEDS["C4"]
    plan = RowBlockPlan(n, world)
    ctx = pa.Context(local_rank)

    # ---- stage 1 on this rank's contigs, then ONE all-gather of the exact count matrix ----
    lo, hi = plan.rows(rank)
    seq, offsets = synthetic.contig_bytes(n, args.length, seed=seed) if world == 1 else \
        synthetic.contig_bytes_range(n, args.length, seed, lo, hi)
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    ctx.count_profiles(d_seq, d_off, args.pattern, "both")          # untimed first call (allocations)
    torch.cuda.synchronize(dev)
    stage1_ms = None
    for _ in range(3):                                               # best of 3, like every other line of the record
        t0 = time.perf_counter()
        my_counts, my_totals = ctx.count_profiles(d_seq, d_off, args.pattern, "both")
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) * 1e3
        stage1_ms = dt if stage1_ms is None else min(stage1_ms, dt)
    allgather_ms = None
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
    if rehearsal and dist is not None:
        counts, totals = first_contact("all_gather_profiles", plan.all_gather_profiles, my_counts.cpu(), my_totals.cpu(), dist)
        counts, totals = counts.to(dev), totals.to(dev)
    elif dist is not None:
        def gather():                     # the path's one exchange and the ranks' first collective: failures surface at the synchronize
            c, t = plan.all_gather_profiles(my_counts, my_totals, dist, force=True)
            torch.cuda.synchronize(dev)
            return c, t
        counts, totals = first_contact("all_gather_profiles", gather)
    else:
        counts, totals = my_counts, my_totals
    if dist is not None:
        torch.cuda.synchronize(dev)
        allgather_ms = (time.perf_counter() - t0) * 1e3
    del d_seq
    dim = counts.shape[1]
    slab, mirrors = plan.allocate(rank, dev, torch.float64)     # this rank's rows x all columns (+ mirror blocks)
    ctx.reserve(n, dim, args.metric)

    def step(want_stats=False, table_path=True):
        return plan.compute(ctx, counts, totals, args.metric, rank, slab, mirrors, want_stats=want_stats,
                            table_path=table_path)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(one)                                   # every rank that took part adds 1
        ranks_seen = int(one.item())
        assert ranks_seen == dist.get_world_size() == world

    # dominant kernel: HIP-event time of the tile kernels on the launch stream (every rank, averaged)
    kms = []
    for _ in range(min(5, max(2, args.steps))):
        st = step(want_stats=True)
        kms.append(st["kernel_ms"])
    kernel_ms = float(np.mean(kms))
    rank_kernel_ms = [kernel_ms]
    complete_rows_ms = None
    if dist is not None:
        g = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(dist.get_world_size())]
        dist.all_gather(g, torch.tensor([kernel_ms], dtype=torch.float64, device=cdev))
        rank_kernel_ms = [float(x.item()) for x in g]

    pairs = n * (n - 1) / 2.0
    ms_per_step = elapsed / args.steps * 1e3
    if rank == 0:
        main_kernel_id = st["kernel_id"]
        rc_folded = bool(st.get("rc_folded", False))
        rank_pairs = float(plan.pair_evaluations(rank))
        # compulsory HBM bytes per unordered pair = two mirrored float64 outputs + the
        # amortised one-time read of both profiles (uint32 counts)
        bytes_per_pair = 2 * 8 + 2 * dim * 4 / (n - 1)
        traffic_all, busy = {}, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic_all = json.load(fh)
        # HBM-side bytes per launch of that kernel from the committed counter passes (FETCH_SIZE x 2 + WRITE_SIZE, separate
        # rocprofv3 --pmc passes, gfx950 correction: MI355X_MICROARCH.md); the key names the workload it was measured on
        traffic = traffic_all.get("%s_n%d_d%d" % (args.metric, n, dim))
        traffic_source = traffic_all.get("_detail", {}).get("source")
        bpath = os.path.join(ROOT, "profiles", "pmc_busy.json")
        if os.path.exists(bpath):
            with open(bpath) as fh:
                busy = json.load(fh)

        def binding(kernel_key):
            """the busiest on-chip resource of that kernel in the committed rocprofv3 --pmc passes"""
            b = busy.get(kernel_key)
            if not b:
                return None
            res = max((k for k in ("valu", "lds", "mfma") if b.get(k) is not None), key=lambda k: b[k])
            return {"resource": res, "busy": b[res], "all": {k: b.get(k) for k in ("valu", "lds", "mfma")}, "source": busy.get("_source")}

        # The committed counter files name the build they were measured on (tools/pmc_busy.py / pmc_traffic.py write the
        # library's po_version(), which carries the hash of its sources): counters of another build are flagged, not trusted.
        from phyloligo_amd import _lib as _polib
        lib_version = _polib.load().po_version().decode()
        lib_hash = lib_version.rsplit("src ", 1)[-1] if "src " in lib_version else None
        stale = []
        if traffic_all.get("_detail", {}).get("src_hash") != lib_hash:
            stale.append("profiles/traffic.json (every roofline.traffic): measured on src %s" % traffic_all.get("_detail", {}).get("src_hash"))
        if busy.get("_src_hash") != lib_hash:
            stale.append("profiles/pmc_busy.json (every roofline.binding): measured on src %s" % busy.get("_src_hash"))

        main_key = {6: "jsd_lut_rows_kernel", 1: "valu_tile_kernel<JSD>", 4: "gram_i8_quad_kernel<f64>", 3: "gram_tile_kernel<f64>",
                    7: "bc_sad_tile_kernel", 2: "valu_tile_kernel<BC>", 8: "pairdot_tile_kernel<KT>", 9: "pairdot_tile_kernel<BC>"}.get(main_kernel_id)
        roof = hbm_roofline(bytes_per_pair * rank_pairs, kernel_ms, traffic=traffic if world == 1 else None, traffic_source=traffic_source,
                            kernel=KERNEL_NAMES.get(main_kernel_id, "tile kernel"), kernel_ms=kernel_ms, bytes_per_pair=bytes_per_pair,
                            binding=binding(main_key),
                            note="nominal roof per north_star; the JSD tile kernels are bound by LDS / vector-ALU issue, "
                                 "not by HBM bytes (one table lookup or logarithm per word and pair for 16 B of output), see DESIGN.md section 3")
        roof["lib_version"] = lib_version
        roof["counters_stale"] = bool(stale)          # true: `traffic` / `binding` below come from rocprof passes over ANOTHER build
        roof["counters_stale_what"] = stale
        general = None
        if args.metric == "JSD" and world == 1:      # the same matrix through the general float64-log kernel only
            gms = [step(want_stats=True, table_path=False) for _ in range(2)]
            g_ms = float(np.mean([x["total_ms"] for x in gms]))
            g_kms = float(np.mean([x["kernel_ms"] for x in gms]))
            general = {"ms": g_ms, "kernel_ms": g_kms, "pairs_per_s": rank_pairs / (g_ms * 1e-3),
                       "roofline": hbm_roofline(bytes_per_pair * rank_pairs, g_kms, binding=binding("valu_tile_kernel<JSD>"),
                                                traffic=traffic_all.get("JSD_general_n%d_d%d" % (n, dim))),
                       "what": "valu_tile_kernel<JSD>: the kernel a ragged real assembly gets (totals differ inside every "
                               "128-record block, or counts above 127)"}
        workload = ("%s: %d synthetic contigs x %d bp (seed %d), pattern %s both strands, -d %s, float64 matrix resident in HBM"
                    % (cfg_name, n, args.length, seed, args.pattern, args.metric))
        result = {
            "metric": "contig-pairs/sec", "value": pairs / (elapsed / args.steps), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "contigs": n, "dim": dim, "pairs": pairs, "sharding": plan.describe(),
                       "stage1_profile_ms": stage1_ms, "matrix_wall_ms": ms_per_step,
                       "rc_folded": rc_folded,      # strand-symmetric profiles summed over one word per {w, rc(w)} orbit
                       "equal_total_table_path": main_kernel_id == 6,
                       "table_path_needs": "every 128-record block to share one word total and counts <= 127 (fixed-length "
                                           "contigs, windows, reads; counts 128 .. 255 take the table's wide layout at ~1.4 x the "
                                           "time); otherwise jsd_general_kernel_only is the rate",
                       "jsd_general_kernel_only": general,
                       "env_knobs": knobs},
            "roofline": roof,
        }
        if general is not None:      # the rate a ragged assembly gets, next to the headline (which needs equal totals)
            result["value_general_kernel"] = general["pairs_per_s"]
            result["value_general_kernel_note"] = ("same matrix through valu_tile_kernel<JSD> only; `value` is the integer-sum table "
                                                    "kernel, which fixed-length synthetic contigs qualify for")
        if world > 1 or dist is not None:
            # the roofline of the dominant kernel on EVERY rank: its own pair evaluations x bytes per pair / its own HIP-event time
            per_rank = []
            for r, ms_r in enumerate(rank_kernel_ms):
                rp = float(plan.pair_evaluations(r))
                per_rank.append(dict(hbm_roofline(bytes_per_pair * rp, ms_r), rank=r, pairs=rp, kernel_ms=ms_r))
            result["roofline_per_rank"] = per_rank
            result["config"]["multi_gpu"] = {
                "ranks_seen": ranks_seen, "backend": "gloo (rehearsal on one GPU)" if rehearsal else "nccl (RCCL)",
                "allgather_ms": allgather_ms, "allgather_bytes": int(counts.numel() * 4 + totals.numel() * 8),
                "complete_rows_ms": complete_rows_ms,
                "kernel_ms_min": min(rank_kernel_ms), "kernel_ms_max": max(rank_kernel_ms), "kernel_ms_per_rank": rank_kernel_ms,
                "pairs_rank0": rank_pairs,
                "environment": __import__("phyloligo_amd.dist", fromlist=["dist_environment"]).dist_environment(),
                "note": "all-gather = the single exchange of the path, outside the timed region like stage 1; "
                        "complete_rows = optional second exchange delivering mirror blocks to row owners, not part of `value`"}
        if world == 1 and args.metric == "JSD" and n == 50000 and not args.no_other_configs:
            # the other single-GPU BASELINE configurations, timed the same way (not part of `value`)
            others = {}
            try:
                def timed(c, t, metric, **kw):
                    best = None
                    for _ in range(3):
                        _, s2 = ctx.pairwise(c, t, metric, out=slab, want_stats=True, **kw)
                        if best is None or s2["total_ms"] < best["total_ms"]:
                            best = s2
                    d_ = c.shape[1]
                    bpp = 16 + 2 * d_ * 4 / (n - 1)
                    return {"ms": best["total_ms"], "kernel_ms": best["kernel_ms"], "pairs_per_s": pairs / (best["total_ms"] * 1e-3),
                            "kernel_id": best["kernel_id"], "kernel": KERNEL_NAMES.get(best["kernel_id"]), "rc_folded": best["rc_folded"],
                            "roofline": hbm_roofline(bpp * pairs, best["kernel_ms"])}
                e = timed(counts, totals, "Eucl")
                e["roofline"]["traffic"] = traffic_all.get("Eucl_n%d_d%d" % (n, dim))
                e["roofline"]["bound"] = "hbm-store"
                e["roofline"]["binding"] = binding("gram_i8_quad_kernel<f64>")
                e["roofline"]["note"] = ("exact int8-MFMA Gram: matrix-core time ~0.1 ms, the kernel is bound by writing 16 B per pair; "
                                         "bare store pattern of this tiling measures 6.1 TB/s (profiles/r01_store_bandwidth.txt)")
                others["C3 Eucl k=4 (default path: exact int8 MFMA)"] = e
                f = timed(counts, totals, "Eucl", table_path=False)
                tf = 2.0 * dim * pairs / (f["kernel_ms"] * 1e-3) / 1e12
                f["roofline"] = {"bound": "mfma-f64", "achieved": tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_MFMA_PEAK_TF,
                                 "flops_per_pair": 2 * dim, "binding": binding("gram_tile_kernel<f64>"),
                                 "traffic": traffic_all.get("Eucl_f64_n%d_d%d" % (n, dim)),
                                 "note": "forced float64 path (v_mfma_f64_16x16x4_f64), the path north_star's MFMA target is quoted on"}
                others["C3 Eucl k=4 (float64 MFMA path, table_path=False)"] = f
                k = timed(counts, totals, "KT")
                k["roofline"]["binding"] = binding("pairdot_tile_kernel<KT>")
                k["roofline"]["traffic"] = traffic_all.get("KT_n%d_d%d" % (n, dim))
                others["C2-size KT k=4"] = k
                seq5, off5 = synthetic.contig_bytes(n, args.length, seed=synthetic.SEEDS["C5"])
                c5, t5 = ctx.count_profiles(torch.from_numpy(seq5).to(dev), torch.from_numpy(off5.astype(np.int64)).to(dev),
                                            "11011011", "both")
                b5 = timed(c5, t5, "BC")
                b5["roofline"]["binding"] = binding("pairdot_tile_kernel<BC>" if b5["kernel_id"] == 9 else "bc_sad_tile_kernel")
                b5["roofline"]["traffic"] = traffic_all.get("BC_n%d_d%d" % (n, c5.shape[1]) if b5["kernel_id"] == 9 else "BC_sad_n%d" % n)
                others["C5 BC pattern 11011011"] = b5
                del c5, t5
                # the float32 matrix of the same assembly - the type of every container (--large memmap / h5py) and of every multi-GPU
                # CLI run; 10 GB instead of 20 (Eucl <= 2.0 ms asked for)
                out32 = torch.empty((n, n), dtype=torch.float32, device=dev)
                f32 = {}
                for m in ("Eucl", "SC", "BC", "KT", "JSD"):
                    best = None
                    for _ in range(3):
                        _, s2 = ctx.pairwise(counts, totals, m, out=out32, dtype="float32", want_stats=True)
                        if best is None or s2["total_ms"] < best["total_ms"]:
                            best = s2
                    bpp32 = 8 + 2 * dim * 4 / (n - 1)
                    f32[m] = {"ms": best["total_ms"], "kernel_ms": best["kernel_ms"], "kernel_id": best["kernel_id"],
                              "kernel": KERNEL_NAMES.get(best["kernel_id"]),
                              "roofline": hbm_roofline(bpp32 * pairs, best["kernel_ms"], bytes_per_pair=bpp32,
                                                       traffic=traffic_all.get({"BC": "BC_sad_f32_n%d_d%d"}.get(m, m + "_f32_n%d_d%d") % (n, dim)))}
                others["C2-size float32 matrix (10 GB)"] = f32
                del out32
            except Exception as exc:             # never let the extras break the headline line
                others["error"] = repr(exc)
            result["config"]["other_configs"] = others
        if world == 1 and args.metric == "JSD" and n == 50000 and not args.no_other_configs:
            # The north star quotes its targets on 200 000 contigs.  Their float64 matrix (320 GB) needs two GPUs (BASELINE config 4,
            # --gpus N), their float32 matrix (160 GB, the container type of --large memmap / h5py) fits one: JSD and Eucl on both its
            # paths at that size, float32 stores (tools/c4_single_gpu.py; parity at this size: tests/test_gpu_c4.py).  Not `value`.
            try:
                free_b, _ = torch.cuda.mem_get_info(dev)
                n4 = 200000
                need = n4 * n4 * 4 + (8 << 30)
                if free_b < need:
                    result["config"]["c4_size_one_gpu_float32"] = {"skipped": "needs %.0f GB of free HBM, %.0f GB are free" % (need / 1e9, free_b / 1e9)}
                else:
                    seq4, off4 = synthetic.contig_bytes(n4, args.length, seed=synthetic.SEEDS["C4"])
                    c4, t4 = ctx.count_profiles(torch.from_numpy(seq4).to(dev), torch.from_numpy(off4.astype(np.int64)).to(dev), args.pattern, "both")
                    del seq4
                    out4 = torch.empty((n4, n4), dtype=torch.float32, device=dev)
                    pairs4 = n4 * (n4 - 1) / 2.0
                    bpp4 = 2 * 4 + 2 * c4.shape[1] * 4 / (n4 - 1)
                    c4rec = {"workload": "BASELINE config 4's assembly on one GPU: %d contigs x %d bp (seed %d), pattern %s both strands, float32 "
                                         "matrix (160 GB) resident in HBM" % (n4, args.length, synthetic.SEEDS["C4"], args.pattern), "pairs": pairs4}
                    for name, metric4, kw in (("JSD", "JSD", {}), ("Eucl_int8", "Eucl", {}), ("Eucl_f64_mfma", "Eucl", {"table_path": False})):
                        best = None
                        for _ in range(2):
                            _, s4 = ctx.pairwise(c4, t4, metric4, dtype="float32", out=out4, want_stats=True, **kw)
                            if best is None or s4["total_ms"] < best["total_ms"]:
                                best = s4
                        rec4 = {"ms": best["total_ms"], "kernel_ms": best["kernel_ms"], "pairs_per_s": pairs4 / (best["total_ms"] * 1e-3),
                                "kernel_id": best["kernel_id"], "roofline": hbm_roofline(bpp4 * pairs4, best["kernel_ms"], bytes_per_pair=bpp4)}
                        if kw:
                            tf4 = 2.0 * c4.shape[1] * pairs4 / (best["kernel_ms"] * 1e-3) / 1e12
                            rec4["roofline"] = {"bound": "mfma-f64", "achieved": tf4, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                                "frac": tf4 / F64_MFMA_PEAK_TF, "flops_per_pair": 2 * int(c4.shape[1])}
                        c4rec[name] = rec4
                    result["config"]["c4_size_one_gpu_float32"] = c4rec
                    del out4, c4, t4
                    torch.cuda.empty_cache()
            except Exception as exc:             # never let the extras break the headline line
                result["config"]["c4_size_one_gpu_float32"] = {"error": repr(exc)}
        if world == 1 and args.metric == "JSD" and not args.no_ragged:
            # What a REAL assembly gets: the same number of contigs with log-normal lengths 1 - 200 kb,
            # four base compositions, N runs, soft-masked stretches and IUPAC codes (synthetic.ragged_assembly; parity of exactly
            # this input: tests/test_gpu_full_size.py::test_ragged_assembly_*).  No two totals alike: every JSD / BC tile is the
            # general kernel's, Eucl / SC take two digit planes.  Not part of `value`.
            try:
                t0 = time.perf_counter()
                rseq, roff = synthetic.ragged_assembly(n, seed=2024)
                gen_s = time.perf_counter() - t0
                d_rseq = torch.from_numpy(rseq).to(dev)
                d_roff = torch.from_numpy(roff.astype(np.int64)).to(dev)
                ctx.count_profiles(d_rseq, d_roff, args.pattern, "both")
                torch.cuda.synchronize(dev)
                r_stage1 = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    rc_, rt_ = ctx.count_profiles(d_rseq, d_roff, args.pattern, "both")
                    torch.cuda.synchronize(dev)
                    dt = (time.perf_counter() - t0) * 1e3
                    r_stage1 = dt if r_stage1 is None else min(r_stage1, dt)
                rdim = rc_.shape[1]
                rbpp = 16 + 2 * rdim * 4 / (n - 1)
                rag = {"workload": "%d contigs, log-normal lengths 1 - 200 kb (%.3f Gb, longest %d), base compositions %s by i %% 4, N runs, "
                                   "lower case, IUPAC codes (phyloligo_amd.synthetic.ragged_assembly, seed 2024), pattern %s both strands"
                                   % (n, rseq.size / 1e9, int(np.diff(roff.astype(np.int64)).max()), list(synthetic.SPECIES), args.pattern),
                       "generate_s": gen_s, "bases": int(rseq.size), "largest_count": int(rc_.max()),
                       "stage1_ms": r_stage1,
                       "stage1_roofline": hbm_roofline(rseq.size + n * rdim * 4 + n * 8, r_stage1,
                                                       note="wall time of po_count_profiles_dev (scan, row zeroing, count_kernel, segment sums; best of 3), algorithmic "
                                                            "bytes = bases + counts + totals (SURVEY 8d)"),
                       "metrics": {}}
                del d_rseq
                for m in ("JSD", "BC", "Eucl", "SC", "KT"):
                    best = None
                    for _ in range(3):
                        _, s2 = ctx.pairwise(rc_, rt_, m, out=slab, want_stats=True)
                        if best is None or s2["total_ms"] < best["total_ms"]:
                            best = s2
                    rag["metrics"][m] = {"ms": best["total_ms"], "kernel_ms": best["kernel_ms"], "prep_ms": best["prep_ms"],
                                         "pairs_per_s": pairs / (best["total_ms"] * 1e-3), "kernel_id": best["kernel_id"],
                                         "kernel": KERNEL_NAMES.get(best["kernel_id"]), "rc_folded": best["rc_folded"],
                                         "roofline": hbm_roofline(rbpp * pairs, best["kernel_ms"],
                                                                  traffic=traffic_all.get("%s_ragged_n%d_d%d" % (m, n, rdim)))}
                rag["metrics"]["JSD"]["roofline"]["binding"] = binding("valu_tile_kernel<JSD>")
                rag["metrics"]["BC"]["roofline"]["binding"] = binding("valu_tile_kernel<BC>")
                rag["metrics"]["KT"]["roofline"]["binding"] = binding("pairdot_tile_kernel<KT>")
                result["config"]["ragged_assembly"] = rag
                result["value_ragged_assembly"] = rag["metrics"]["JSD"]["pairs_per_s"]
                result["value_ragged_assembly_note"] = ("pairs/s of -d JSD on the ragged assembly (valu_tile_kernel<JSD> owns every tile): the "
                                                        "rate a real input gets; `value` needs equal word totals")
                del rc_, rt_
            except Exception as exc:             # never let the extras break the headline line
                result["config"]["ragged_assembly"] = {"error": repr(exc)}
        if world == 1 and args.metric == "JSD" and not args.no_path_lines:
            # "stage-1 time, H2D, D2H and .mat writing are reported as separate lines, never hidden" - none of them
            # is part of `value`.  Measured on THIS workload (the C2 assembly when the defaults are used).
            try:
                result["config"]["path_lines"] = path_lines(ctx, seq, offsets, counts, totals, n, dim, args.metric, args.pattern, dev, stage1_ms, ms_per_step)
            except Exception as exc:             # never let the extras break the headline line
                result["config"]["path_lines"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import phyloligo_oracle as po
            freq = po.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
            result["cpu_baseline"] = cpu_baseline(freq, args.metric)
            try:     # the same BASELINE config 1 job on the GPU, from sequence bytes in host memory to the float64 matrix in host memory
                seqs = po.synthetic_contigs(1000, 2000, seed=1001)
                s1 = np.frombuffer(b"".join(seqs), dtype=np.uint8)
                o1 = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint64)
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    c1, t1 = ctx.count_profiles(s1, o1, "1111", "both")
                    t_prof = time.perf_counter() - t0
                    m1 = ctx.pairwise(c1, t1, "Eucl")
                    t_all = time.perf_counter() - t0
                    if best is None or t_all < best[1]:
                        best = (t_prof, t_all)
                result["cpu_baseline"]["c1_full"]["same_job_on_the_gpu"] = {
                    "profiles_s": best[0], "distances_s": best[1] - best[0], "whole_s": best[1],
                    "note": "host pointers in, host float64 matrix out (H2D, kernels, D2H), best of 3; matrix %s" % (m1.shape,)}
            except Exception as exc:             # never let the extras break the headline line
                result["cpu_baseline"]["c1_full"]["same_job_on_the_gpu"] = {"error": repr(exc)}
    # The optional second exchange (mirror blocks to their row owners: point to point over RCCL, never run on more than one GPU
    # so far) comes LAST, behind a watchdog: if it does not complete, rank 0 still prints the record - whose `value` does not
    # depend on it - with the failure named, and every rank leaves.
    if dist is not None and world > 1 and not args.no_complete_rows:
        import threading
        finished = threading.Event()
        emit_lock, emitted = threading.Lock(), [False]

        def emit(error=None):                    # the one JSON line, whoever gets here first (watchdog thread or main thread)
            with emit_lock:
                if rank == 0 and not emitted[0]:
                    emitted[0] = True
                    if error is not None:
                        result["config"]["multi_gpu"]["complete_rows_error"] = error
                    print(json.dumps(result), flush=True)

        def watchdog():
            if finished.wait(args.complete_rows_timeout):
                return
            emit("no completion within %g s; the ranks left without it" % args.complete_rows_timeout)
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if rehearsal:                        # gloo moves host tensors; the slab stays on the device
                plan.complete_rows(rank, slab, [None if m is None else m.cpu() for m in mirrors], dist, stage_device="cpu")
            else:
                plan.complete_rows(rank, slab, mirrors, dist)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) * 1e3], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if rank == 0:
                result["config"]["multi_gpu"]["complete_rows_ms"] = float(t.item())
        except Exception as exc:                 # noqa: BLE001 - the record goes out whatever this optional step does
            finished.set()
            emit(repr(exc)[:400])
            os._exit(0)                          # a failed collective leaves the group unusable: no barrier, no destroy
        finished.set()
        emit()
    elif rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
