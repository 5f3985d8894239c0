"""one-off: writing the 10 GB container into (a) a new file, (b) an existing file opened with O_TRUNC, (c) an existing file of the
same size without truncation; po_pwrite_rows with 8 threads from a 1 GB host buffer."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from phyloligo_amd import phyloligo as P
n = 50000
buf = np.random.default_rng(0).random((5000, n), dtype=np.float32)
path = "/tmp/ow.f32"


def write(fd):
    for r0 in range(0, n, 5000):
        P._pwrite_rows(fd, buf, r0, 0, n, threads=8)


for mode in ("new", "trunc", "keep", "keep", "trunc", "unlink_new"):
    if mode == "new" and os.path.exists(path):
        os.remove(path)
    t0 = time.time()
    if mode == "unlink_new":
        os.remove(path)
    t_rm = time.time() - t0
    flags = os.O_RDWR | os.O_CREAT | (os.O_TRUNC if mode == "trunc" else 0)
    fd = os.open(path, flags, 0o666)
    t_open = time.time() - t0
    if mode != "keep" or os.fstat(fd).st_size != n * n * 4:
        P._reserve_file(fd, n * n * 4)
    t_res = time.time() - t0
    write(fd)
    os.close(fd)
    print("%-10s remove %.3f  open %.3f  reserve %.3f  write %.3f  total %.3f s" % (mode, t_rm, t_open - t_rm, t_res - t_open, time.time() - t0 - t_res, time.time() - t0), flush=True)
os.remove(path)
