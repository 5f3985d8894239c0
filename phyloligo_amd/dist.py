"""Sharding of the path over the GPUs of one node (one process per GPU, torch.distributed / RCCL).

Reference analogue: even row slices handed to joblib workers with the whole frequency matrix
visible to each (gen_even_slices, /root/reference/phylopackage/bin/phyloligo.py:424,516).

Here contigs are split into contiguous, tile-aligned row blocks R_0..R_{G-1}.  Stage 1 runs on each
rank's own contigs, ONE all-gather moves the exact integer count matrix (the only exchange the path
needs), then the upper triangle of the block grid is dealt out like a round-robin tournament so that
every unordered pair of records is evaluated exactly once in the whole job:

  rank g computes   R_g x R_g                      (triangular, mirrored inside its own slab)
                    R_g x R_{(g+s) mod G}          s = 1 .. ceil(G/2)-1   (+ the transposed block,
                                                   kept in a mirror buffer owned by rank g+s)
  and, for even G,  half of R_g x R_{g+G/2}        (the two partners split that block by rows of the
                                                   lower-numbered one)

Every rank does (N/G)^2 * G/2 pair evaluations: the same total as one GPU using symmetry, perfectly
balanced, no collective inside the compute.  After it, every matrix entry (i,j) and (j,i) is resident
in HBM on the rank that evaluated the pair; `complete_rows` is the optional second exchange (point to
point over xGMI) that sends each mirror buffer to the rank whose row slab it belongs to, for
consumers that want row-complete slabs (e.g. writing the .mat).
"""
import json
import os
import sys

import numpy as np

# what the ranks of a multi-GPU run inherit and what decides whether RCCL comes up at all: reported with every N > 1 record and
# with every first-contact failure (no process group of more than one GPU has ever run this code - see DESIGN section 6)
_ENV_PREFIXES = ("HSA_", "NCCL_", "RCCL_", "HIP_", "ROCR_", "GPU_", "MASTER_", "TORCH_NCCL_", "PO_")
_ENV_NAMES = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "OMP_NUM_THREADS", "CUDA_VISIBLE_DEVICES")


def dist_environment():
    """The environment a rank runs under, as it goes into the bench record (`config.multi_gpu.environment`) and into the
    error line of a failed first contact: the variables that steer HSA / RCCL / the launcher, the torch and RCCL versions.
    HSA_ENABLE_IPC_MODE_LEGACY is named even when unset: launch.spawn_ranks sets it to 0 (dmabuf IPC) unless the caller
    has set it, on the strength of a ONE-rank RCCL test - the first run on several GPUs is where that gets verified."""
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith(_ENV_PREFIXES) or k in _ENV_NAMES}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "(unset)")
    info = {"env": env}
    try:
        import torch
        info["torch"] = torch.__version__
        info["hip"] = getattr(torch.version, "hip", None)
        try:
            info["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as exc:                        # no GPU build / no device: say so instead of failing the report
            info["rccl"] = "unavailable (%s)" % type(exc).__name__
        info["devices_visible"] = torch.cuda.device_count()
    except Exception as exc:
        info["torch"] = "unavailable (%s)" % type(exc).__name__
    return info


def first_contact(stage, fn, *args, **kwargs):
    """Run one step of bringing the ranks together (init_process_group, the first collective).  If it raises, the rank says so in
    ONE JSON line - on stdout for rank 0, the line the driver parses; on stderr for the others - with the stage, the error and the
    environment it ran under, and exits with status 3: torch.distributed.run then takes the other ranks down and
    launch.spawn_ranks hands the status on.  A run that cannot start must not look like a hang or a Python traceback lottery."""
    try:
        return fn(*args, **kwargs)
    except BaseException as exc:                         # (KeyboardInterrupt / SystemExit included: still a failed start)
        rank = int(os.environ.get("RANK", "0"))
        line = {"error": "%s: %s" % (type(exc).__name__, exc), "stage": stage, "rank": rank,
                "world_size": int(os.environ.get("WORLD_SIZE", "1"))}
        line.update(dist_environment())
        out = sys.stdout if rank == 0 else sys.stderr
        out.write(json.dumps(line) + "\n")
        out.flush()
        if rank != 0:                                    # the launcher takes every rank down when the first one exits: leave
            import time                                  # rank 0 - which usually fails for the same reason, but may still be
            time.sleep(15.0)                             # importing torch on a loaded host - the time to get its line out
        sys.exit(3)


class RowBlockPlan:
    def __init__(self, n, world, align=128):
        self.n, self.world = int(n), int(world)
        per = -(-self.n // self.world)
        per = -(-per // align) * align            # tile aligned so no two ranks share a tile row
        self.align = align
        self.bounds = [min(self.n, r * per) for r in range(self.world + 1)]
        self.bounds[-1] = self.n

    def rows(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def describe(self):
        return "tournament over %d row block(s), bounds %s" % (self.world, self.bounds if self.world <= 8 else "...")

    # ---- the single exchange of the path: exact count matrix to every rank ------------------------
    def all_gather_profiles(self, my_counts, my_totals, dist, force=False):
        """counts[n, dim] int32 / totals[n] int64 on every rank.  force=True issues the collective even in a
        one-rank group (bench.py's PO_BENCH_FORCE_DIST and the GPU tests: RCCL exercised on a one-GPU box)."""
        import torch
        if dist is None or (self.world == 1 and not force):
            return my_counts, my_totals
        dim = my_counts.shape[1]
        counts = torch.empty((self.n, dim), dtype=my_counts.dtype, device=my_counts.device)
        totals = torch.empty((self.n,), dtype=my_totals.dtype, device=my_totals.device)
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        if len(set(sizes)) == 1:
            dist.all_gather_into_tensor(counts, my_counts.contiguous())
            dist.all_gather_into_tensor(totals, my_totals.contiguous())
            return counts, totals
        # ragged last block(s): gather equal-size padded blocks, keep the valid rows (works on
        # every backend; the padding is at most one block of the count matrix)
        per = max(sizes)
        pc = torch.zeros((per, dim), dtype=my_counts.dtype, device=my_counts.device)
        pt = torch.zeros((per,), dtype=my_totals.dtype, device=my_totals.device)
        pc[:my_counts.shape[0]] = my_counts
        pt[:my_totals.shape[0]] = my_totals
        gc = torch.empty((self.world * per, dim), dtype=my_counts.dtype, device=my_counts.device)
        gt = torch.empty((self.world * per,), dtype=my_totals.dtype, device=my_totals.device)
        dist.all_gather_into_tensor(gc, pc)
        dist.all_gather_into_tensor(gt, pt)
        for r in range(self.world):
            lo, hi = self.bounds[r], self.bounds[r + 1]
            counts[lo:hi] = gc[r * per:r * per + (hi - lo)]
            totals[lo:hi] = gt[r * per:r * per + (hi - lo)]
        return counts, totals

    # ---- tournament schedule ---------------------------------------------------------------------------
    def _half(self, a):
        """split point of R_a for the half blocks of an even world (tile aligned)"""
        lo, hi = self.rows(a)
        mid = lo + (-(-((hi - lo) // 2) // self.align)) * self.align
        return min(mid, hi)

    def work(self, rank):
        """List of (rows, cols, kind, peer) this rank evaluates.  kind: 'diag' (triangular, no peer),
        'full' / 'half' (rectangular; the transposed block belongs to `peer`'s slab)."""
        G, g = self.world, rank
        lo, hi = self.rows(g)
        items = [((lo, hi), (lo, hi), "diag", None)]
        for s in range(1, (G + 1) // 2):
            p = (g + s) % G
            items.append(((lo, hi), self.rows(p), "full", p))
        if G % 2 == 0 and G > 1:
            p = (g + G // 2) % G
            a = min(g, p)
            mid = self._half(a)
            if g == a:      # lower partner: first half of its own rows x all of the partner's columns
                items.append(((lo, mid), self.rows(p), "half", p))
            else:           # upper partner: all of its rows x second half of the partner's rows (as columns)
                items.append(((lo, hi), (mid, self.rows(a)[1]), "half", p))
        return [it for it in items if it[0][1] > it[0][0] and it[1][1] > it[1][0]]

    def pair_evaluations(self, rank):
        tot = 0
        for (r0, r1), (c0, c1), kind, _ in self.work(rank):
            tot += (r1 - r0) * (r1 - r0 + 1) // 2 if kind == "diag" else (r1 - r0) * (c1 - c0)
        return tot

    def allocate(self, rank, device, dtype):
        """Row slab [rows_g, n] and one mirror buffer per rectangular block."""
        import torch
        lo, hi = self.rows(rank)
        # Rows start on 128-byte boundaries whatever n is (row stride rounded up to 32 elements; the slab is a view of its first n
        # columns): the tile kernels' 16-byte stores need 16-byte aligned rows, and HBM takes row pieces that start on a line
        # boundary ~25 % faster than pieces that straddle one (tools/ubench/write_bw_f32.hip: 5.4 against 4.0 - 5.1 TB/s)
        pad = lambda c: -(-c // 32) * 32
        slab = torch.empty((hi - lo, pad(self.n)), dtype=dtype, device=device)[:, :self.n]
        mirrors = []
        for (r0, r1), (c0, c1), kind, peer in self.work(rank):
            mirrors.append(None if kind == "diag" else torch.empty((c1 - c0, pad(r1 - r0)), dtype=dtype, device=device)[:, :r1 - r0])
        return slab, mirrors

    def blocks(self, rank, slab, mirrors):
        """Block list for Context.pairwise_blocks."""
        lo, _ = self.rows(rank)
        out = []
        for ((r0, r1), (c0, c1), kind, peer), m in zip(self.work(rank), mirrors):
            view = slab[r0 - lo:r1 - lo, c0:c1]
            if kind == "diag":
                out.append({"rows": (r0, r1), "cols": (c0, c1), "out": view, "triangular": True})
            else:
                out.append({"rows": (r0, r1), "cols": (c0, c1), "out": view, "mirror": m})
        return out

    def compute(self, ctx, counts, totals, metric, rank, slab, mirrors, want_stats=False, table_path=True):
        return ctx.pairwise_blocks(counts, totals, metric, self.blocks(rank, slab, mirrors), dtype=slab.dtype,
                                   want_stats=want_stats, table_path=table_path)

    def complete_rows(self, rank, slab, mirrors, dist, chunk_bytes=256 << 20, stage_device=None):
        """Second, optional exchange: send every mirror buffer to the rank whose slab it completes and
        place the received ones.  Point to point (RCCL send/recv over xGMI).  There is at most one message per
        ordered pair of ranks; every message is cut into row chunks of <= chunk_bytes and chunk k of all messages
        is in flight at once, so the receive side needs one bounded staging buffer per peer (not a second copy of
        its mirrors: at 200 000 contigs on 2 GPUs those are 40 GB next to a 160 GB slab).
        stage_device: where the receive staging lives (default: with the slab).  The rehearsal on one GPU moves host tensors
        over gloo - mirrors given as host tensors, stage_device="cpu" - and still places them into the slab on the device."""
        import torch
        lo, hi = self.rows(rank)
        if dist is None or self.world == 1:
            return slab
        esz = slab.element_size()

        def chunking(rows, cols):
            per = max(1, chunk_bytes // max(1, cols * esz))
            return per, -(-rows // per)

        sends, recvs, steps = [], [], 0
        for src in range(self.world):
            for idx, ((r0, r1), (c0, c1), kind, peer) in enumerate(self.work(src)):
                if kind == "diag":
                    continue
                per, nch = chunking(c1 - c0, r1 - r0)          # the message is the transposed block [c1-c0, r1-r0]
                steps = max(steps, nch)
                if src == rank:
                    sends.append((mirrors[idx], peer, per, nch))
                if peer == rank:                                 # rows [c0,c1) of my slab, columns [r0,r1)
                    stage = torch.empty((min(per, c1 - c0), r1 - r0), dtype=slab.dtype, device=stage_device or slab.device)
                    recvs.append((stage, src, per, nch, c0 - lo, c1 - lo, r0, r1))
        for k in range(steps):
            ops, placed = [], []
            for m, peer, per, nch in sends:
                if k < nch:
                    piece = m[k * per:min(m.shape[0], (k + 1) * per)]
                    # (mirror buffers keep their rows on 128-byte boundaries: a block whose width is not a multiple of 32 is a
                    #  strided view, and a message has to be contiguous - one packed copy of the chunk, <= chunk_bytes)
                    ops.append(dist.P2POp(dist.isend, piece if piece.is_contiguous() else piece.contiguous(), peer))
            for stage, src, per, nch, a0, a1, b0, b1 in recvs:
                if k < nch:
                    rows = min(a1 - a0, (k + 1) * per) - k * per
                    ops.append(dist.P2POp(dist.irecv, stage[:rows], src))
                    placed.append((stage[:rows], a0 + k * per, a0 + k * per + rows, b0, b1))
            if ops:                               # a rank with an empty row block has nothing to move
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for buf, a0, a1, b0, b1 in placed:
                slab[a0:a1, b0:b1] = buf
        return slab


def assemble_virtual(plan, slabs, mirrors_by_rank):
    """Single-process stand-in for complete_rows (tests, one GPU): place every mirror buffer into the
    slab of the rank that owns those rows and return the full matrix."""
    import torch
    for g in range(plan.world):
        for ((r0, r1), (c0, c1), kind, peer), m in zip(plan.work(g), mirrors_by_rank[g]):
            if kind != "diag":
                plo, _ = plan.rows(peer)
                slabs[peer][c0 - plo:c1 - plo, r0:r1] = m
    return torch.cat(slabs, dim=0)
