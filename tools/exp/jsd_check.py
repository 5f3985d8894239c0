"""JSD at C2 size: kernel time (best and all of 4 launches) and a checksum of corners + random entries + a symmetry test of the
result, so that variants of the table kernel can be compared for equal output (run once per library through tools/exp/ab.sh)."""
import sys, hashlib
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dtype = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=dtype, device="cuda")
idx = torch.from_numpy(np.random.default_rng(1).integers(0, n, size=(2, 200000))).cuda()
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
out.fill_(-7.0)
ms = []
for _ in range(4):
    _, st = ctx.pairwise(c, t, "JSD", out=out, want_stats=True)
    ms.append(st["kernel_ms"])
h = hashlib.sha1(out[:1500, :1500].cpu().numpy().tobytes() + out[n - 700:, n - 900:].cpu().numpy().tobytes() + out[idx[0], idx[1]].cpu().numpy().tobytes()).hexdigest()[:16]
sym = bool(torch.equal(out[:3000, :3000], out[:3000, :3000].T)) and bool(torch.equal(out[n - 3000:, :3000], out[:3000, n - 3000:].T))
print("JSD kernel best %6.2f ms  (%s)  checksum %s  symmetric corners %s  unwritten %d" % (min(ms), " ".join("%.2f" % m for m in ms), h, sym, int((out[idx[0], idx[1]] == -7.0).sum())), flush=True)
