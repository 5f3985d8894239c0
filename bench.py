#!/usr/bin/env python3
"""Headline benchmark: contig-pairs/sec of the all-by-all distance matrix (BASELINE.json).

A step = one pass of stage 2 (counts resident in HBM -> every matrix entry of this rank's shard
resident in HBM, float64) over one synthetic assembly.  N=1: BASELINE config 2, 50,000 contigs x
2 kb, k=4, both strands, -d JSD.  N>1: one process per GPU (torch.distributed, RCCL), the count
matrix is all-gathered once, then the upper triangle of the block grid is dealt out tournament-style
(phyloligo_amd/dist.py) so that every pair is evaluated once, with no further exchange inside the timed
region; the assembly grows as 50,000*sqrt(N) contigs so that the pairs per GPU stay fixed (weak scaling).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline_jsd(freq, metric, budget_rows):
    """The oracle's per-pair Python path (= the reference's joblib path: one metric call per
    pair under sklearn.pairwise_distances) timed on a bounded slab of rows x all columns."""
    import joblib
    from oracle import phyloligo_oracle as po
    cores = min(os.cpu_count() or 1, 16)
    rows_per = max(1, budget_rows // cores)
    slabs = [list(range(c * rows_per, (c + 1) * rows_per)) for c in range(cores)]
    t0 = time.perf_counter()
    joblib.Parallel(n_jobs=cores)(joblib.delayed(po.pairwise_rows)(freq, metric, rows) for rows in slabs)
    dt = time.perf_counter() - t0
    evaluated = cores * rows_per * freq.shape[0]
    return {"value": evaluated / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d rows x %d columns = %d metric calls of oracle.%s (per-pair numpy path of "
                      "phylodist.py) in %.1f s over %d processes" % (cores * rows_per, freq.shape[0], evaluated, metric, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--contigs", type=int, default=50000, help="contigs at 1 GPU (BASELINE config 2)")
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("--metric", default="JSD", choices=["Eucl", "JSD", "KT", "BC", "SC"])
    ap.add_argument("--pattern", default="1111")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 extras in config.other_configs")
    args = ap.parse_args()

    import torch
    import phyloligo_amd as pa
    from phyloligo_amd import synthetic
    from phyloligo_amd.dist import RowBlockPlan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PO_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo collectives -- lets the N>1 code path be run on
    # a one-GPU box (numbers are meaningless then); the real run is one rank per GPU over RCCL.
    rehearsal = os.environ.get("PO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # PO_BENCH_FORCE_DIST=1: take the torch.distributed path even with one rank (exercises the RCCL calls on a
    # one-GPU box: process group, all_gather_into_tensor, barrier, all_reduce)
    if world > 1 or os.environ.get("PO_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)

    n = args.contigs if world == 1 else int(round(args.contigs * math.sqrt(world) / (256 * world))) * 256 * world
    seed = synthetic.SEEDS["C2"]
    plan = RowBlockPlan(n, world)
    ctx = pa.Context(local_rank)

    # ---- stage 1 on this rank's contigs, then ONE all-gather of the exact count matrix ----
    lo, hi = plan.rows(rank)
    seq, offsets = synthetic.contig_bytes(n, args.length, seed=seed) if world == 1 else \
        synthetic.contig_bytes_range(n, args.length, seed, lo, hi)
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    my_counts, my_totals = ctx.count_profiles(d_seq, d_off, args.pattern, "both")
    torch.cuda.synchronize(dev)
    stage1_ms = (time.perf_counter() - t0) * 1e3
    if rehearsal and dist is not None:
        counts, totals = plan.all_gather_profiles(my_counts.cpu(), my_totals.cpu(), dist)
        counts, totals = counts.to(dev), totals.to(dev)
    else:
        counts, totals = plan.all_gather_profiles(my_counts, my_totals, dist)
    dim = counts.shape[1]
    rows = hi - lo
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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs = n * (n - 1) / 2.0
    ms_per_step = elapsed / args.steps * 1e3
    result = None
    if rank == 0:
        # dominant kernel: HIP-event time of the tile kernel on the launch stream, averaged
        kms = []
        for _ in range(min(5, max(2, args.steps))):
            st = step(want_stats=True)
            kms.append(st["kernel_ms"])
        kernel_ms = float(np.mean(kms))
        main_kernel_id = st["kernel_id"]
        rc_folded = bool(st.get("rc_folded", False))
        rank_pairs = float(plan.pair_evaluations(rank))
        # compulsory HBM bytes per unordered pair = two mirrored float64 outputs + the
        # amortised one-time read of both profiles (uint32 counts)
        bytes_per_pair = 2 * 8 + 2 * dim * 4 / (n - 1)
        algo_bytes = bytes_per_pair * rank_pairs
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get("%s_n%d_d%d" % (args.metric, n, dim))
        general = None
        if args.metric == "JSD":      # the same matrix through the general float64-log kernel only
            gms = []
            for _ in range(2):
                gms.append(step(want_stats=True, table_path=False)["total_ms"])
            general = {"ms": float(np.mean(gms)), "pairs_per_s": rank_pairs / (float(np.mean(gms)) * 1e-3)}
        kernel_names = {1: "valu_tile_kernel<JSD>", 2: "valu_tile_kernel<BC>", 3: "gram_tile_kernel (f64 MFMA)",
                        5: "kt_tile_kernel", 6: "jsd_lut_tile_kernel (equal-total record blocks) + valu_tile_kernel<JSD> (rest)"}
        result = {
            "metric": "contig-pairs/sec", "value": pairs / (elapsed / args.steps), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d synthetic contigs x %d bp (seed %d), pattern %s both strands, -d %s, "
                                   "float64 matrix resident in HBM" % (n, args.length, seed, args.pattern, args.metric),
                       "contigs": n, "dim": dim, "pairs": pairs, "sharding": plan.describe(),
                       "stage1_profile_ms": stage1_ms, "matrix_wall_ms": ms_per_step,
                       "rc_folded": rc_folded,      # strand-symmetric profiles summed over one word per {w, rc(w)} orbit
                       "jsd_general_kernel_only": general},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_names.get(main_kernel_id, "tile kernel"),
                         "kernel_ms": kernel_ms, "bytes_per_pair": bytes_per_pair,
                         "note": "nominal roof per north_star; the tile kernels are bound by vector-ALU / LDS issue, "
                                 "not by HBM bytes, see DESIGN.md section 3"},
        }
        if world == 1 and args.metric == "JSD" and args.contigs == 50000 and not args.no_other_configs:
            # the other single-GPU BASELINE configurations, timed the same way (not part of `value`): C3 = the same
            # assembly with -d Eucl, C5 = pattern 11011011 (D = 4096) with -d BC
            others = {}
            try:
                def timed(c, t, metric):
                    best = None
                    for _ in range(3):
                        _, st = ctx.pairwise(c, t, metric, out=slab, want_stats=True)
                        if best is None or st["total_ms"] < best["total_ms"]:
                            best = st
                    return {"ms": best["total_ms"], "kernel_ms": best["kernel_ms"], "pairs_per_s": pairs / (best["total_ms"] * 1e-3),
                            "kernel_id": best["kernel_id"], "rc_folded": best["rc_folded"]}
                others["C3 Eucl k=4"] = timed(counts, totals, "Eucl")
                seq5, off5 = synthetic.contig_bytes(n, args.length, seed=synthetic.SEEDS["C5"])
                c5, t5 = ctx.count_profiles(torch.from_numpy(seq5).to(dev), torch.from_numpy(off5.astype(np.int64)).to(dev),
                                            "11011011", "both")
                others["C5 BC pattern 11011011"] = timed(c5, t5, "BC")
                del c5, t5
            except Exception as exc:             # never let the extras break the headline line
                others["error"] = repr(exc)
            result["config"]["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            from oracle import phyloligo_oracle as po
            freq = po.counts_to_frequencies(counts.cpu().numpy().astype(np.int64), totals.cpu().numpy())
            result["cpu_baseline"] = cpu_baseline_jsd(freq, args.metric, budget_rows=128)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
