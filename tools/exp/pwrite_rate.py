"""How fast can the float32 container be written on this box?  pwrite of a 512 MB host buffer, repeated to 4 GB, into a
fresh file with 4..32 threads (page cache), with and without preallocation, O_DIRECT if the filesystem takes it, and
into /dev/shm for the memory-speed bound.  Usage: python tools/exp/pwrite_rate.py [dir]"""
import concurrent.futures as cf
import mmap
import os
import sys
import time

import numpy as np

where = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
BLOCK = 512 << 20
TOTAL = 4 << 30
buf = np.random.default_rng(1).integers(0, 255, BLOCK, dtype=np.uint8)
flat = memoryview(buf)


def put(fd, view, offset):
    done = 0
    while done < len(view):
        done += os.pwrite(fd, view[done:], offset + done)


def run(path, threads, piece_bytes=None, prealloc=False, direct=False):
    flags = os.O_RDWR | os.O_CREAT | os.O_TRUNC | (os.O_DIRECT if direct else 0)
    try:
        fd = os.open(path, flags, 0o666)
    except OSError as e:
        return "open failed: %s" % e
    try:
        t0 = time.perf_counter()
        if prealloc:
            os.posix_fallocate(fd, 0, TOTAL)
        else:
            os.ftruncate(fd, TOTAL)
        t_alloc = time.perf_counter() - t0
        src = flat
        if direct:                                   # O_DIRECT wants page-aligned memory
            m = mmap.mmap(-1, BLOCK)
            m.write(buf.tobytes())
            src = memoryview(m)
        with cf.ThreadPoolExecutor(max_workers=threads) as pool:
            for lo in range(0, TOTAL, BLOCK):
                piece = piece_bytes or -(-BLOCK // threads)
                piece = -(-piece // 4096) * 4096
                fs = [pool.submit(put, fd, src[a:a + piece], lo + a) for a in range(0, BLOCK, piece)]
                for f in fs:
                    f.result()
        dt = time.perf_counter() - t0
        return "%5.2f GB/s (%.2f s, allocation %.2f s)" % (TOTAL / dt / 1e9, dt, t_alloc)
    except OSError as e:
        return "failed: %s" % e
    finally:
        os.close(fd)
        try:
            os.unlink(path)
        except OSError:
            pass


print("cpus usable:", len(os.sched_getaffinity(0)), " dir:", where, flush=True)
os.system("df -T %s | tail -1" % where)
for thr in (4, 8, 12, 16, 24, 32):
    print("page cache, %2d threads, one piece per thread : %s" % (thr, run(os.path.join(where, "po_pw.bin"), thr)), flush=True)
for thr in (8, 16):
    print("page cache, %2d threads, 4 MiB pieces         : %s" % (thr, run(os.path.join(where, "po_pw.bin"), thr, piece_bytes=4 << 20)), flush=True)
    print("page cache, %2d threads, preallocated         : %s" % (thr, run(os.path.join(where, "po_pw.bin"), thr, prealloc=True)), flush=True)
    print("O_DIRECT,   %2d threads                       : %s" % (thr, run(os.path.join(where, "po_pw.bin"), thr, direct=True)), flush=True)
for thr in (8, 16):
    print("/dev/shm,   %2d threads                       : %s" % (thr, run("/dev/shm/po_pw.bin", thr)), flush=True)
