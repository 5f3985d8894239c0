"""Where a cold `python -m phyloligo_amd ... --large memmap` spends its time at C2 size (fresh process, phases timed)."""
import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, '.')
import numpy as np
t_np = time.perf_counter()
import torch
t_torch = time.perf_counter()
import phyloligo_amd as pa
from phyloligo_amd import phyloligo as P, synthetic
t_imp = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
tmp = os.environ.get("TMPDIR", "/tmp")
fa = os.path.join(tmp, "cold_%d.fa" % n)
if not os.path.exists(fa):
    seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
    open(fa, "wb").write(synthetic.fasta_bytes(seq, off))
t0 = time.perf_counter()
ctx = P._context()
t_ctx = time.perf_counter()
ing = P.read_fasta_device(fa)
torch.cuda.synchronize()
t_ing = time.perf_counter()
d_seq, d_off, titles = ing
c, t = ctx.count_profiles(d_seq, d_off, "1111", "both")
torch.cuda.synchronize()
t_cnt = time.perf_counter()
f = ctx.frequencies(c, t)
torch.cuda.synchronize()
t_f = time.perf_counter()
fh = f.cpu().numpy(); ch = c.cpu().numpy(); th = t.cpu().numpy()
t_d2h = time.perf_counter()
print("imports: numpy %.2f s, torch %.2f s, package %.2f s" % (t_np - t00, t_torch - t_np, t_imp - t_torch))
print("context %.1f ms | file -> HBM -> records %.1f ms | count %.1f ms | frequencies %.1f ms | profiles to host %.1f ms"
      % ((t_ctx - t0) * 1e3, (t_ing - t_ctx) * 1e3, (t_cnt - t_ing) * 1e3, (t_f - t_cnt) * 1e3, (t_d2h - t_f) * 1e3))
t1 = time.perf_counter()
freq, _ = P.compute_frequencies("hip", "memmap", fa, "1111", "both", 250, 4, tmp)
print("compute_frequencies (second pass over the same file) %.1f ms" % ((time.perf_counter() - t1) * 1e3))
out = os.path.join(tmp, "cold.f32")
t1 = time.perf_counter()
P.compute_distances("hip", "memmap", freq, None, out, "JSD", 4, 250, tmp)
t2 = time.perf_counter()
print("compute_distances memmap %.1f ms (%.1f GB/s of container)" % ((t2 - t1) * 1e3, n * n * 4 / (t2 - t1) / 1e9))
t1 = time.perf_counter()
P.compute_distances("hip", "memmap", freq, None, out, "JSD", 4, 250, tmp)
t2 = time.perf_counter()
print("compute_distances memmap again %.1f ms" % ((t2 - t1) * 1e3))
os.remove(out)
