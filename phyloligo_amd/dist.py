"""Sharding of the path over the GPUs of one node (one process per GPU).

Reference analogue: even row slices handed to joblib workers with the whole frequency matrix
visible to each (gen_even_slices, /root/reference/phylopackage/bin/phyloligo.py:424,516).
Here: contigs are split into contiguous row blocks; stage 1 runs on each rank's own contigs,
ONE all-gather moves the exact integer count matrix (RCCL over xGMI), then every rank computes
rows [lo,hi) x all columns of the matrix with no further exchange.
"""
import numpy as np


class RowBlockPlan:
    def __init__(self, n, world, align=128):
        self.n, self.world = int(n), int(world)
        per = -(-self.n // self.world)
        per = -(-per // align) * align            # tile aligned so no two ranks share a tile row
        self.bounds = [min(self.n, r * per) for r in range(self.world + 1)]
        self.bounds[-1] = self.n

    def rows(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def describe(self):
        return "row blocks x all columns, %d rank(s), bounds %s" % (self.world, self.bounds if self.world <= 8 else "...")

    def all_gather_profiles(self, my_counts, my_totals, dist):
        """counts[n, dim] int32 / totals[n] int64 on every rank.  Row blocks are ragged, so the
        exchange is an all_gather into per-rank views of the final tensors."""
        import torch
        if dist is None or self.world == 1:
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
