"""Time the sliding-window scan (Kount mirror) on a synthetic genome: stage 1 over overlapping ranges."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import kount
ctx = pa.Context(0)
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(3)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=L, dtype=np.uint8)]
wins = kount.record_windows(L, 5000, 500)
begins = np.array([w[0] for w in wins], dtype=np.uint64); ends = np.minimum(L, begins + 5000).astype(np.uint64)
print("genome %.0f Mb, %d windows of 5000 / step 500 (%.2f GB of window bytes)" % (L / 1e6, len(wins), (ends - begins).sum() / 1e9))
proto_c, proto_t = ctx.count_profiles(seq, np.array([0, L], dtype=np.uint64), "1111", "both")
proto = proto_c[0] / float(proto_t[0])
for it in range(3):
    t = time.perf_counter(); counts, totals = ctx.count_profiles_ranges(seq, begins, ends, "1111", "both"); t1 = time.perf_counter() - t
    t = time.perf_counter(); d = ctx.profile_distances(counts, totals, proto, "JSD"); t2 = time.perf_counter() - t
    print("  windows counted in %.1f ms (host buffers: H2D %d MB + D2H %d MB included), distances in %.2f ms; mean JSDx1000 %.3f"
          % (t1 * 1e3, L >> 20, counts.nbytes >> 20, t2 * 1e3, d.mean() * 1000))
