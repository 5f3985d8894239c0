"""one-off: where the time of compute_frequencies() goes on the C2 FASTA (97 MB, 50 000 records): every phase of the device ingest,
timed cold (first call of a fresh process after the context exists) and warm (third call)."""
import os, sys, tempfile, time, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import phyloligo_amd as pa
from phyloligo_amd import synthetic, api, _lib
from phyloligo_amd import phyloligo as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR") or "/tmp")
fa = os.path.join(tmp, "assembly.fa")
with open(fa, "wb") as fh:
    fh.write(synthetic.fasta_bytes(seq, off))
size = os.path.getsize(fa)
torch.cuda.init(); torch.zeros(1, device="cuda")
lib = _lib.load()


def phases(ctx):
    out = []
    def mark(name, t0):
        torch.cuda.synchronize(); t1 = time.perf_counter(); out.append((name, (t1 - t0) * 1e3)); return t1
    t = time.perf_counter()
    buf = np.empty(size, dtype=np.uint8); t = mark("np.empty", t)
    api.check(lib.po_file_read(fa.encode(), api._np_ptr(buf), size)); t = mark("po_file_read", t)
    raw = torch.from_numpy(buf).to("cuda"); t = mark("H2D of the file", t)
    nrec, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
    ctx._use_torch_stream()
    api.check(lib.po_fasta_scan_dev(ctx._h, raw.data_ptr(), size, ctypes.byref(nrec), ctypes.byref(nbytes))); t = mark("scan", t)
    s = torch.empty(((nbytes.value + 15) // 16 * 16,), dtype=torch.uint8, device="cuda")
    o = torch.zeros((nrec.value + 1,), dtype=torch.int64, device="cuda")
    tb = torch.empty((nrec.value,), dtype=torch.int64, device="cuda"); te = torch.empty_like(tb); t = mark("allocations", t)
    api.check(lib.po_fasta_extract_dev(ctx._h, raw.data_ptr(), size, s.data_ptr(), o.data_ptr(), tb.data_ptr(), te.data_ptr())); t = mark("extract", t)
    titles = api._LineTitles(buf, tb.cpu().numpy(), te.cpu().numpy()); t = mark("title spans D2H", t)
    c, tot = ctx.count_profiles(s[:nbytes.value], o, "1111", "both"); t = mark("stage 1", t)
    f = ctx.frequencies(c, tot); t = mark("frequencies (device)", t)
    fh = f.cpu().numpy(); t = mark("frequencies D2H", t)
    ch = c.cpu().numpy().view(np.uint32); th = tot.cpu().numpy().view(np.uint64); t = mark("counts + totals D2H", t)
    return out


ctx = pa.Context(0)
for rnd in range(3):
    t0 = time.perf_counter()
    ph = phases(ctx)
    tot = (time.perf_counter() - t0) * 1e3
    print("round %d: %.1f ms   " % (rnd, tot) + "  ".join("%s %.1f" % p for p in ph), flush=True)
for rnd in range(3):
    t0 = time.perf_counter()
    freq, _ = P.compute_frequencies("hip", "memmap", fa, "1111", "both", 250, 4, tmp)
    print("compute_frequencies() call %d: %.1f ms" % (rnd, (time.perf_counter() - t0) * 1e3), flush=True)
# the bench's own sequence: a SECOND context (phyloligo.py keeps its own), created on first use
import importlib
P._ctx = None
t0 = time.perf_counter(); c2 = pa.Context(0); t1 = time.perf_counter()
print("second Context(): %.1f ms" % ((t1 - t0) * 1e3), flush=True)
c2.close()
with open(fa, "wb") as fh:
    fh.write(synthetic.fasta_bytes(seq, off))
P._ctx = None
torch.cuda.empty_cache()
for rnd in range(3):
    t0 = time.perf_counter()
    freq, _ = P.compute_frequencies("hip", "memmap", fa, "1111", "both", 250, 4, tmp)
    print("fresh module context, empty torch cache: compute_frequencies() call %d: %.1f ms" % (rnd, (time.perf_counter() - t0) * 1e3), flush=True)
os.remove(fa)
