"""KT (C2 size) and BC (C5): kernel time, and a checksum of a corner + random entries of the result so that variants can be
compared for equal output (run once per library through tools/exp/ab.sh)."""
import sys, hashlib
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
idx = torch.from_numpy(np.random.default_rng(1).integers(0, n, size=(2, 200000))).cuda()
for metric, pattern, seed in (("KT", "1111", 50001), ("BC", "11011011", 50005)):
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")
    best = 1e9
    out.fill_(-7.0)
    for _ in range(3):
        _, st = ctx.pairwise(c, t, metric, out=out, want_stats=True)
        best = min(best, st["kernel_ms"])
    h = hashlib.sha1(out[:1500, :1500].cpu().numpy().tobytes() + out[n - 700:, n - 900:].cpu().numpy().tobytes() + out[idx[0], idx[1]].cpu().numpy().tobytes()).hexdigest()[:16]
    sym = bool(torch.equal(out[:3000, :3000], out[:3000, :3000].T))
    print("%s kernel %7.2f ms  id %d  checksum %s  symmetric corner %s  unwritten %d" % (metric, best, st["kernel_id"], h, sym, int((out[idx[0], idx[1]] == -7.0).sum())), flush=True)
