"""Host ingest at C2 scale: FASTA bytes -> (sequence bytes, offsets) -> device -> profiles, step by step."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, '.')
import phyloligo_amd as pa
from phyloligo_amd import synthetic, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ctx = pa.Context(0)
seq, off = synthetic.contig_bytes(n, 2000, seed=50001)
fa = synthetic.fasta_bytes(seq, off)
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "asm.fa")
    open(path, "wb").write(fa)
    for it in range(3):
        t0 = time.perf_counter()
        data = np.fromfile(path, dtype=np.uint8)
        t1 = time.perf_counter()
        s, o, names = api.fasta_index(data)
        t2 = time.perf_counter()
        ds = torch.from_numpy(s).cuda(); do = torch.from_numpy(o.astype(np.int64)).cuda(); torch.cuda.synchronize()
        t3 = time.perf_counter()
        c, t = ctx.count_profiles(ds, do, "1111", "both"); torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("N=%d FASTA %d MB: read %.1f ms | parse %.1f ms (%.2f GB/s) | H2D %.1f ms | profiles %.2f ms"
              % (n, len(fa) >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(fa) / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
    # the same through the on-device parser: parallel pread -> H2D of the raw file -> po_fasta_scan_dev / extract_dev
    from phyloligo_amd import _lib
    import ctypes
    lib = _lib.load()
    for it in range(3):
        t0 = time.perf_counter()
        buf = np.empty(os.path.getsize(path), dtype=np.uint8)
        _lib.check(lib.po_file_read(path.encode(), ctypes.c_void_p(buf.ctypes.data), buf.size))
        t1 = time.perf_counter()
        raw = torch.from_numpy(buf).cuda(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ds, do, names = api.fasta_index_dev(ctx, path); torch.cuda.synchronize()
        t3 = time.perf_counter()
        c, t = ctx.count_profiles(ds, do, "1111", "both"); torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("device parser: read (parallel pread) %.1f ms | H2D raw %.1f ms | fasta_index_dev all in (read + H2D + parse) %.1f ms | profiles %.2f ms"
              % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
