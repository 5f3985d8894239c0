import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic, _lib
lib = _lib.load()
n = 50000
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
c, t = ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), "1111", "both")
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
for _ in range(2):
    _, st = ctx.pairwise(c, t, "JSD", out=out, want_stats=True)
print("JSD kernel %.2f ms" % st["kernel_ms"])
if hasattr(lib, "po_debug_jsd_lut_stamps"):
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert lib.po_debug_jsd_lut_stamps(ctypes.c_void_p(buf.ctypes.data), buf.size) == 0
    s = buf.reshape(4096, 2, 4).astype(np.float64)
    for g in (0, 1):
        x = s[:, g]
        ghz = x[:, 0] / (x[:, 1] * 10.0)
        print("wave %d: loop %.0f cycles for %d words = %.0f cycles per word (32 lookups per lane), %.1f us, clock %.2f GHz"
              % (7 * g, x[:, 0].mean(), x[0, 2], (x[:, 0] / x[:, 2]).mean(), (x[:, 1] * 0.01).mean(), ghz.mean()))
