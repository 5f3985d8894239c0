"""one-off: the FASTA parsers (device: po_fasta_scan_dev / po_fasta_extract_dev; host: po_fasta_scan / po_fasta_extract) on a
file of more than 4 GiB with one record of 3.3 GB wrapped at 80 columns and one of 1.2 GB on a single line."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import phyloligo_amd as pa
from phyloligo_amd import api, phyloligo as P

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/big.fa"
rng = np.random.default_rng(3)
alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
specs = [("a small one", 1_000_003, 60), ("big wrapped", 3_300_000_017, 80), ("c single line", 1_200_000_000, 0), ("d", 5_001, 70)]
seqs = []
t0 = time.time()
with open(path, "wb") as fh:
    for title, L, width in specs:
        s = alphabet[rng.integers(0, 4, size=L, dtype=np.uint8)]
        seqs.append(s)
        fh.write(b">" + title.encode() + b"\n")
        if width == 0:
            s.tofile(fh); fh.write(b"\n")
        else:
            full = (L // width) * width
            body = np.empty((L // width, width + 1), dtype=np.uint8)
            body[:, :width] = s[:full].reshape(-1, width)
            body[:, width] = 10
            body.tofile(fh)
            if L > full:
                s[full:].tofile(fh); fh.write(b"\n")
            del body
print("wrote %.2f GB in %.0f s" % (os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
want_off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
ctx = P._context()
t0 = time.time()
d_seq, d_off, titles = api.fasta_index_dev(ctx, path)
torch.cuda.synchronize()
print("device parser: %.2f s, %d records, titles %s" % (time.time() - t0, d_off.numel() - 1, titles), flush=True)
ok_off = np.array_equal(d_off.cpu().numpy(), want_off)
ok_seq = True
for i, s in enumerate(seqs):
    a, b = int(want_off[i]), int(want_off[i + 1])
    for lo in range(a, b, 1 << 30):
        hi = min(b, lo + (1 << 30))
        ok_seq = ok_seq and bool(torch.equal(d_seq[lo:hi].cpu(), torch.from_numpy(s[lo - a:hi - a])))
print("device parser: offsets", ok_off, " sequence bytes", ok_seq, flush=True)
del d_seq
t0 = time.time()
h_seq, h_off, h_titles = P.read_fasta(path)
print("host parser: %.2f s" % (time.time() - t0), flush=True)
ok_h = np.array_equal(np.asarray(h_off, dtype=np.int64), want_off) and list(h_titles) == list(titles)
for i, s in enumerate(seqs):
    ok_h = ok_h and np.array_equal(np.asarray(h_seq[want_off[i]:want_off[i + 1]]), s)
print("host parser: offsets, titles and bytes", ok_h, flush=True)
os.remove(path)
print("ALL OK" if (ok_off and ok_seq and ok_h) else "MISMATCH")
