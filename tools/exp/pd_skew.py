"""K-stage skew between the workgroups of pairdot_tile_kernel that share an L2 (is the TCC hit rate of 64 %
explained by co-resident tiles drifting apart in K?).  Needs the diagnostic variant (tools/exp/pairdot_skew_stamps.patch built with
-DPD_SKEW into tools/exp/variants/libskew.so; run with PO_ALLOW_VARIANT=1 PO_LIB_PATH=tools/exp/variants/libskew.so).
Wave 0 of every workgroup stamps s_memrealtime at stage 0, every eighth of the stages, the end of the Gram loop and the end of the
kernel, with its tile and its XCC id.  For sample times across the launch: the stage every Gram-phase workgroup of an XCD is at
(linear between stamps), and for every (workgroup, operand block) whether another workgroup of the SAME XCD reading the SAME block is
at most L stages ahead (it pulled those stage pieces through the L2 a moment ago: a hit) - the lock-step hit rate the tile order was
designed for - against the distance distribution of the sharers."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
lib = _lib.load()
if not hasattr(lib, "po_debug_pairdot_skew"):
    print("no skew stamps in this build"); sys.exit(0)
lib.po_debug_pairdot_skew.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ctx = pa.Context(0)
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
SLOTS, BLOCKS = 16, 32768


def report(name, kernel_ms):
    buf = np.zeros(BLOCKS * SLOTS, dtype=np.uint64)
    assert lib.po_debug_pairdot_skew(ctypes.c_void_p(buf.ctypes.data), buf.size) == 0
    s = buf.reshape(BLOCKS, SLOTS)
    s = s[s[:, 9] > 0]
    t = s[:, :10].astype(np.float64) * 10e-9                       # seconds (100 MHz)
    t0 = t[:, 0].min()
    t = t - t0
    ti, tj = (s[:, 11] >> 32).astype(np.int64), (s[:, 11] & 0xFFFFFFFF).astype(np.int64)
    nst, xcc = s[:, 12].astype(np.float64), s[:, 13].astype(np.int64)
    nb = len(s)
    print("%s: kernel %.2f ms, %d workgroups stamped, %d stages each; XCC ids seen %s; blockIdx %% 8 == XCC id for %.1f %% of them"
          % (name, kernel_ms, nb, int(nst[0]), sorted(set(xcc.tolist())), 100.0 * np.mean((np.arange(nb) % 8) == xcc) if nb == len(buf.reshape(BLOCKS, SLOTS)[:nb]) else -1))
    gram = t[:, 8] - t[:, 0]
    epi = t[:, 9] - t[:, 8]
    print("  Gram phase %.1f us (p10 %.1f, p90 %.1f), epilogue + stores %.1f us (p10 %.1f, p90 %.1f); launch span %.2f ms"
          % (gram.mean() * 1e6, np.percentile(gram, 10) * 1e6, np.percentile(gram, 90) * 1e6, epi.mean() * 1e6,
             np.percentile(epi, 10) * 1e6, np.percentile(epi, 90) * 1e6, t[:, 9].max() * 1e3))
    # stage position of every workgroup at time x: piecewise linear through the 9 stamps (stage k * nst / 8 at stamp k)
    knots = np.arange(9)[None, :] * (nst[:, None] / 8.0)
    samples = np.linspace(0.05, 0.95, 37) * t[:, 9].max()
    dist_hist = np.zeros(7)
    edges = [1, 2, 4, 8, 16, 32]
    lead = {L: [0, 0] for L in (1, 2, 3, 4, 8)}
    alive_counts = []
    for x in samples:
        on = (t[:, 0] <= x) & (x < t[:, 8])
        alive_counts.append(on.sum())
        for c in range(8):
            idx = np.flatnonzero(on & (xcc == c))
            if idx.size < 2:
                continue
            pos = np.array([np.interp(x, t[i, :9], knots[i]) for i in idx])
            blocks = [(ti[i], tj[i]) for i in idx]
            for a in range(idx.size):
                for blk in set(blocks[a]):                         # its row block and its column block (one on the diagonal)
                    sharers = [b for b in range(idx.size) if b != a and blk in blocks[b]]
                    ahead = [pos[b] - pos[a] for b in sharers]
                    for L in lead:
                        lead[L][1] += 1
                        lead[L][0] += any(0.0 <= d <= L for d in ahead)
                    if ahead:
                        d = min(abs(v) for v in ahead)
                        dist_hist[np.searchsorted(edges, d, side="right")] += 1
                    else:
                        dist_hist[6] += 0                          # nobody shares it right now (counted in `lead` as a miss)
    print("  workgroups in their Gram phase at a sample time: mean %.0f of %d slots" % (np.mean(alive_counts), 256))
    tot = dist_hist.sum()
    print("  nearest sharer of an operand block on the same XCD, |stage distance|: " +
          "  ".join("%s: %.1f %%" % (lab, 100 * v / tot) for lab, v in zip(("<1", "1-2", "2-4", "4-8", "8-16", "16-32", ">=32"), dist_hist)))
    for L in sorted(lead):
        print("  (workgroup, block) requests with a sharer 0..%d stages AHEAD on the same XCD: %.1f %%" % (L, 100.0 * lead[L][0] / max(1, lead[L][1])))


def profiles(pattern, seed):
    seq, off = synthetic.contig_bytes(n, 2000, seed=seed)
    return ctx.count_profiles(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), pattern, "both")


c, t = profiles("1111", 50001)
_, st = ctx.pairwise(c, t, "KT", out=out, want_stats=True)
buf = np.zeros(BLOCKS * SLOTS, dtype=np.uint64); lib.po_debug_pairdot_skew(ctypes.c_void_p(buf.ctypes.data), buf.size)   # drop the warm-up's stamps
_, st = ctx.pairwise(c, t, "KT", out=out, want_stats=True)
report("KT (C2 size, K = 9 472 folded)", st["kernel_ms"])
c, t = profiles("11011011", 50005)
_, st = ctx.pairwise(c, t, "BC", out=out, want_stats=True)
lib.po_debug_pairdot_skew(ctypes.c_void_p(buf.ctypes.data), buf.size)
_, st = ctx.pairwise(c, t, "BC", out=out, want_stats=True)
report("BC (C5, thermometer planes)", st["kernel_ms"])
