import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import phyloligo_amd as pa
from test_gpu_rc_fold import contigs_ragged, pack
ctx = pa.Context(0)
contigs = contigs_ragged(150, 77)
seq, off = pack(contigs)
for pattern in sys.argv[1:] or ["11011011"]:
    counts, totals = ctx.count_profiles(seq, off, pattern, "both")
    for metric in ("JSD", "BC"):
        for tp in (True, False):
            a, st = ctx.pairwise(counts, totals, metric, want_stats=True, table_path=tp)
            b, st0 = ctx.pairwise(counts, totals, metric, want_stats=True, rc_fold=False, table_path=tp)
            d = np.abs(a - b)
            d[np.isnan(d)] = 0
            i, j = np.unravel_index(np.argmax(d), d.shape)
            print(pattern, metric, tp, st["rc_folded"], "max abs diff", d.max(), "at", (i, j), a[i, j], b[i, j], "totals", totals[i], totals[j],
                  "bad rows", np.unique(np.where(d > 1e-12)[0])[:20])
