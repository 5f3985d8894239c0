"""PCIe-inclusive numbers: the host-pointer entry points (numpy in, numpy out) and the CLI end to end."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic, phyloligo as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
pairs = n * (n - 1) / 2
for metric in ("JSD", "Eucl"):
    t2 = 1e9
    for it in range(3):
        t = time.perf_counter(); counts, totals = ctx.count_profiles(seq, off, "1111", "both"); t1 = time.perf_counter() - t
        m = None                                     # free the previous result outside the timed region (munmap of 7 GB)
        t = time.perf_counter(); m = ctx.pairwise(counts, totals, metric); t2 = min(t2, time.perf_counter() - t)
    print("host-pointer ABI N=%d %s: po_count_profiles %.1f ms (H2D %d MB, D2H %d MB) | po_pairwise %.1f ms (D2H %.1f GB) -> %.3e pairs/s PCIe-inclusive"
          % (n, metric, t1 * 1e3, seq.nbytes >> 20, counts.nbytes >> 20, t2 * 1e3, m.nbytes / 1e9, pairs / t2), flush=True)
    del m
with tempfile.TemporaryDirectory() as tmp:
    fa = os.path.join(tmp, "asm.fa")
    open(fa, "wb").write(synthetic.fasta_bytes(seq, off))
    for large, out in (("memmap", "out.f32"), ("None", "out.mat")):
        nn = n if large == "memmap" else min(n, 5000)
        if large == "None" and nn != n:
            s2, o2 = synthetic.contig_bytes(nn, 2000, seed=50001)
            open(fa, "wb").write(synthetic.fasta_bytes(s2, o2))
        t = time.perf_counter()
        P.main(["-i", fa, "-k", "4", "-d", "JSD", "--method", "joblib", "--large", large, "-o", os.path.join(tmp, out)])
        dt = time.perf_counter() - t
        print("CLI N=%d --large %s: %.2f s end to end (FASTA %d MB -> %s %.2f GB)" % (nn, large, dt, os.path.getsize(fa) >> 20, out, os.path.getsize(os.path.join(tmp, out)) / 1e9), flush=True)
