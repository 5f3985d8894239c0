"""Same number of pairs as a triangular block and as a rectangular block with mirror (the two block kinds of the
multi-GPU work lists): po_pairwise_blocks_dev time for each."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
metric = sys.argv[1] if len(sys.argv) > 1 else "JSD"
ctx = pa.Context(0)
n = 75264
seq, off = synthetic.contig_bytes(n, 2000, seed=5)
dseq = torch.from_numpy(seq).cuda(); doff = torch.from_numpy(off.astype(np.int64)).cuda()
counts, totals = ctx.count_profiles(dseq, doff, "1111", "both")
m = 25088
tri_n = 35456            # 35456^2/2 ~ 25088^2
tri = torch.empty((tri_n, tri_n), dtype=torch.float64, device="cuda")
rect = torch.empty((m, m), dtype=torch.float64, device="cuda")
mir = torch.empty((m, m), dtype=torch.float64, device="cuda")
cases = {
    "triangular %d" % tri_n: [{"rows": (0, tri_n), "cols": (0, tri_n), "out": tri, "triangular": True}],
    "rect %dx%d + mirror" % (m, m): [{"rows": (0, m), "cols": (m, 2 * m), "out": rect, "mirror": mir}],
    "rect far columns": [{"rows": (0, m), "cols": (2 * m, 3 * m), "out": rect, "mirror": mir}],
}
for name, blocks in cases.items():
    best = None
    for _ in range(4):
        st = ctx.pairwise_blocks(counts, totals, metric, blocks, want_stats=True)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    print("%-28s prep %.2f ms kernel %.2f ms  %.3e pairs/s (kernel)" % (name, best["prep_ms"], best["kernel_ms"], best["pairs"] / (best["kernel_ms"] * 1e-3)), flush=True)
