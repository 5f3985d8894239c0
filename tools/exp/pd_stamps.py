"""Phase breakdown of pairdot_tile_kernel from in-kernel s_memtime stamps (diagnostic variant build -DPD_STAMPS only;
run through tools/exp/ab.sh so that PO_LIB_PATH points at it).  KT at C2 size and BC at C5."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
lib = _lib.load()
if not hasattr(lib, "po_debug_pairdot_stamps"):
    print("no stamps in this build"); sys.exit(0)
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
SLOTS, BLOCKS = 16, 4096


def report(name):
    buf = np.zeros(BLOCKS * 2 * SLOTS, dtype=np.uint64)
    rc = lib.po_debug_pairdot_stamps(ctypes.c_void_p(buf.ctypes.data), buf.size)
    assert rc == 0
    s = buf.reshape(BLOCKS, 2, SLOTS).astype(np.float64)
    s = s[s[:, 0, 9] > 0]                                   # blocks that ran
    for g in (0, 1):
        x = s[:, g]
        gram, epi = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1]
        ghz = (x[:, 2] - x[:, 0]) / ((x[:, 4] - x[:, 3]) * 10.0)      # memrealtime ticks are 10 ns
        st = x[:, 9]
        print("%s wave %d: blocks %d stages %d | gram %.0f cyc (%.1f us) epilogue+stores %.0f cyc (%.1f us) | per stage: wait-own-dma %.0f barrier %.0f issue %.0f compute %.0f cyc | clock %.2f GHz"
              % (name, 4 * g, len(x), st[0], gram.mean(), gram.mean() / ghz.mean() / 1e3, epi.mean(), epi.mean() / ghz.mean() / 1e3,
                 (x[:, 5] / st).mean(), (x[:, 6] / st).mean(), (x[:, 7] / st).mean(), (x[:, 8] / st).mean(), ghz.mean()), flush=True)


def profiles(pattern, seed):
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    return ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")


c, t = profiles("1111", 50001)
for _ in range(2):
    _, st = ctx.pairwise(c, t, "KT", out=out, want_stats=True)
print("KT kernel %.2f ms" % st["kernel_ms"])
report("KT")
c, t = profiles("11011011", 50005)
for _ in range(2):
    _, st = ctx.pairwise(c, t, "BC", out=out, want_stats=True)
print("BC kernel %.2f ms" % st["kernel_ms"])
report("BC")
