"""Diagnostic build of the JSD table kernel (-DJL_STAMPS, tools/exp/build_variant.sh): s_memtime stamps of every wave's items
(means per wave) - word loop, reading the next item's number, epilogue + store issue, drain of the stores (an added s_waitcnt vmcnt(0))."""
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
    x = buf.reshape(4096, 8).astype(np.float64)
    x = x[x[:, 6] > 0]
    items = x[:, 6].copy()
    print("items per wave: mean %.1f min %d max %d" % (items.mean(), items.min(), items.max()))
    tot = x[:, :6].sum(axis=0)
    print("share of all wave time: word loop %.1f %%, next number %.1f %%, epilogue + store issue %.1f %%, drain %.1f %%; per item (all waves): loop %.0f, epilogue %.0f cycles"
          % (100 * tot[0] / tot[4], 100 * tot[1] / tot[4], 100 * tot[2] / tot[4], 100 * tot[3] / tot[4], tot[0] / items.sum(), tot[2] / items.sum()))
    x[:, :6] /= items[:, None]
    ghz = (x[:, 4] / (x[:, 5] * 10.0)).mean()
    names = ("word loop", "next item's number", "epilogue + store issue", "drain of the stores (vmcnt(0))", "whole item")
    for i, nm in enumerate(names):
        v = x[:, i]
        print("%-34s mean %9.0f cycles = %6.2f us   (p10 %9.0f, median %9.0f, p90 %9.0f)" % (nm, v.mean(), v.mean() / ghz / 1e3, np.percentile(v, 10), np.median(v), np.percentile(v, 90)))
    print("clock %.2f GHz, %d waves reported" % (ghz, len(x)))
