"""one-off: Bray-Curtis at C2 size (packed SAD kernel), float64 and float32, and at C5 (pattern 11011011: FP4 thermometer Gram): kernel_ms best of 4"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic
ctx = pa.Context(0)
n = 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
dseq, doff = torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
outs = {"float64": torch.empty((n, n), dtype=torch.float64, device="cuda"), "float32": torch.empty((n, n), dtype=torch.float32, device="cuda")}
for pattern in ("1111", "11011011"):
    c, t = ctx.count_profiles(dseq, doff, pattern, "both")
    row = []
    for name, out in outs.items():
        ks = []
        for _ in range(4):
            _, st = ctx.pairwise(c, t, "BC", out=out, want_stats=True, dtype=name)
            ks.append(st["kernel_ms"])
        row.append("%s kernel %6.2f ms" % (name, min(ks)))
    print("BC pattern %-9s (id %d)  %s   checksum %.17g" % (pattern, st["kernel_id"], "   ".join(row), float(outs["float64"][:2000, :2000].sum())), flush=True)
