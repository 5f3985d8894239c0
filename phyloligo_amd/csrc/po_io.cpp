// Host-side formats either side of the path: FASTA ingest and the text .mat writer.
//
// FASTA: the reference iterates Bio.SeqIO.parse(genome, "fasta") and uses str(record.seq)
// (/root/reference/phylopackage/bin/phyloligo.py:869).  Biopython is third-party and absent from
// the reference tree; the semantics restated here are those of its SimpleFastaParser: a record
// opens at a line whose first byte is '>', the title is the rest of that line right-stripped,
// the sequence is the following lines each right-stripped and joined, with every ' ' and '\r'
// removed.  Blank lines before the first record are skipped; any other text there is an error.
//
// .mat: numpy.savetxt(path, m, delimiter="\t") (/root/reference/phylopackage/bin/phyloligo.py:1061,
// :1066) = "%.18e" per value, '\t' between columns, '\n' after every row, nothing else.
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <functional>
#include <memory>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "phyloligo_amd.h"
#include "po_host.h"

#include <sched.h>

void po_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// (declared in po_host.h)
// Host threads worth starting: the CPUs this process may run on (affinity mask), cut down to the cgroup-v2 CPU quota
// when there is one (a GPU box shows every hardware thread of the host but gives a job a share of them), at most `cap`.
unsigned po_host_threads(unsigned cap) {
    static const unsigned usable = []() -> unsigned {
        unsigned n = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
        if (n == 0) n = std::thread::hardware_concurrency();
        if (n == 0) n = 4;
        if (FILE* fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32];
            double period = 0.0;
            if (fscanf(fh, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) {
                const double cpus = atof(q) / period;
                if (cpus >= 1.0 && cpus < (double)n) n = (unsigned)(cpus + 0.5);
            }
            fclose(fh);
        }
        return n;
    }();
    return usable < cap ? usable : (cap ? cap : 1u);
}

namespace {
// work(t) for t in [0, n) on n host threads, and wait.  When the system refuses a thread (std::system_error) the shares of the
// threads that could not be started run on the calling thread: no exception leaves through the C ABI and no joinable
// std::thread is destroyed (that would be std::terminate).
template <class F>
void po_run_threads(unsigned n, F&& work) {
    if (n <= 1) {
        if (n) work(0u);
        return;
    }
    std::vector<std::thread> pool;
    unsigned started = 0;
    try {
        pool.reserve(n);
        for (; started < n; ++started) pool.emplace_back(std::ref(work), started);
    } catch (...) {
    }
    for (unsigned t = started; t < n; ++t) work(t);
    for (auto& th : pool) th.join();
}
}  // namespace

// An extern "C" function whose body allocates: std::bad_alloc (or anything else) becomes a status, not an abort of the caller.
#define PO_C_GUARD_BEGIN try {
#define PO_C_GUARD_END(name)                                                                     \
    } catch (const std::bad_alloc&) {                                                            \
        po_set_error(name ": out of host memory");                                               \
        return PO_ENOMEM;                                                                        \
    } catch (const std::exception& e) {                                                          \
        po_set_error(name ": %s", e.what());                                                     \
        return PO_EIO;                                                                           \
    }

namespace {

inline bool is_py_space(uint8_t c) {  // bytes.rstrip() / str.rstrip() default set for ASCII
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c;
}

// Walks the lines of data[begin, end) (begin is a line start).  With seq_out == NULL only counts.
// rec_base / out_base: records and sequence bytes that precede this segment.  Returns 0 or PO_EIO.
int fasta_walk(const uint8_t* data, uint64_t begin, uint64_t end, bool in_record, uint64_t rec_base, uint64_t out_base,
               uint8_t* seq_out, uint64_t* offsets_out, uint64_t* title_begin, uint64_t* title_end,
               uint64_t* n_records, uint64_t* seq_bytes) {
    uint64_t pos = begin, nrec = rec_base, nout = out_base;
    while (pos < end) {
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(data + pos, '\n', end - pos));
        const uint64_t line_end = nl ? (uint64_t)(nl - data) : end;   // exclusive, without the '\n'
        uint64_t e = line_end;
        while (e > pos && is_py_space(data[e - 1])) --e;              // rstrip
        if (data[pos] == '>' && line_end > pos) {
            if (offsets_out) offsets_out[nrec] = nout;
            if (title_begin) title_begin[nrec] = pos + 1;
            if (title_end) title_end[nrec] = (e > pos + 1) ? e : pos + 1;
            ++nrec;
            in_record = true;
        } else if (!in_record) {
            if (e > pos) {
                po_set_error("FASTA input does not start with '>' (byte %llu)", (unsigned long long)pos);
                return PO_EIO;
            }
        } else if (seq_out) {
            for (uint64_t i = pos; i < e; ++i) {
                const uint8_t c = data[i];
                if (c == ' ' || c == '\r') continue;
                seq_out[nout++] = c;
            }
        } else {
            for (uint64_t i = pos; i < e; ++i) nout += (data[i] != ' ' && data[i] != '\r');
        }
        pos = nl ? line_end + 1 : end;
    }
    if (n_records) *n_records = nrec - rec_base;
    if (seq_bytes) *seq_bytes = nout - out_base;
    return PO_OK;
}

// The file is cut at line starts into one segment per host thread.  Only the text before the first record
// depends on what came earlier (it must be blank), so that prelude is walked first, on its own; every later
// segment starts inside a record.  Pass 1 counts records and sequence bytes per segment, a prefix sum places
// them, pass 2 (extract only) writes.  Same result as one sequential walk.
int fasta_parallel(const uint8_t* data, uint64_t len, uint8_t* seq_out, uint64_t* offsets_out, uint64_t* title_begin,
                   uint64_t* title_end, uint64_t* n_records, uint64_t* seq_bytes) {
    // prelude: up to the first line that starts with '>'
    uint64_t first = 0;
    while (first < len) {
        if (data[first] == '>') break;
        const uint8_t* nl = static_cast<const uint8_t*>(memchr(data + first, '\n', len - first));
        if (!nl) { first = len; break; }
        first = (uint64_t)(nl - data) + 1;
    }
    int rc = fasta_walk(data, 0, first, false, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;

    uint64_t nseg = po_host_threads(32);
    const uint64_t body = len - first;
    if (body / (4u << 20) + 1 < nseg) nseg = body / (4u << 20) + 1;   // at least 4 MiB per thread
    std::vector<uint64_t> cut(nseg + 1, len);
    cut[0] = first;
    for (uint64_t s = 1; s < nseg; ++s) {
        uint64_t p = first + body / nseg * s;
        if (p < cut[s - 1]) p = cut[s - 1];
        const uint8_t* nl = p < len ? static_cast<const uint8_t*>(memchr(data + p, '\n', len - p)) : nullptr;
        cut[s] = nl ? (uint64_t)(nl - data) + 1 : len;
    }
    std::vector<uint64_t> nrec(nseg, 0), nout(nseg, 0);
    std::vector<int> rcs(nseg, PO_OK);
    auto run = [&](const std::function<void(uint64_t)>& body_fn) {
        po_run_threads((unsigned)nseg, [&](unsigned s) { body_fn(s); });
    };
    run([&](uint64_t s) {
        rcs[s] = fasta_walk(data, cut[s], cut[s + 1], true, 0, 0, nullptr, nullptr, nullptr, nullptr, &nrec[s], &nout[s]);
    });
    uint64_t rec_total = 0, out_total = 0;
    std::vector<uint64_t> rec_base(nseg), out_base(nseg);
    for (uint64_t s = 0; s < nseg; ++s) {
        rec_base[s] = rec_total; out_base[s] = out_total;
        rec_total += nrec[s]; out_total += nout[s];
    }
    if (n_records) *n_records = rec_total;
    if (seq_bytes) *seq_bytes = out_total;
    if (!offsets_out) return PO_OK;
    run([&](uint64_t s) {
        rcs[s] = fasta_walk(data, cut[s], cut[s + 1], true, rec_base[s], out_base[s], seq_out, offsets_out, title_begin,
                            title_end, nullptr, nullptr);
    });
    offsets_out[rec_total] = out_total;
    return PO_OK;
}

}  // namespace

extern "C" int po_fasta_scan(const uint8_t* data, uint64_t len, uint64_t* n_records, uint64_t* seq_bytes) {
    if ((!data && len) || !n_records || !seq_bytes) {
        po_set_error("po_fasta_scan: NULL argument");
        return PO_EINVAL;
    }
    PO_C_GUARD_BEGIN
    return fasta_parallel(data, len, nullptr, nullptr, nullptr, nullptr, n_records, seq_bytes);
    PO_C_GUARD_END("po_fasta_scan")
}

extern "C" int po_fasta_extract(const uint8_t* data, uint64_t len, uint8_t* seq_out, uint64_t* offsets_out,
                                uint64_t* title_begin, uint64_t* title_end) {
    if ((!data && len) || !offsets_out) {
        po_set_error("po_fasta_extract: NULL argument");
        return PO_EINVAL;
    }
    PO_C_GUARD_BEGIN
    return fasta_parallel(data, len, seq_out, offsets_out, title_begin, title_end, nullptr, nullptr);
    PO_C_GUARD_END("po_fasta_extract")
}

// The first `len` bytes of a file into buf, read by the host threads the job may use (pread of disjoint ranges):
// the raw FASTA bytes on their way to HBM for po_fasta_scan_dev / po_fasta_extract_dev.
extern "C" int po_file_read(const char* path, uint8_t* buf, uint64_t len) {
    if (!path || (!buf && len)) { po_set_error("po_file_read: NULL argument"); return PO_EINVAL; }
    if (len == 0) return PO_OK;
    unsigned nthr = po_host_threads(16);
    if (len / (8u << 20) + 1 < nthr) nthr = (unsigned)(len / (8u << 20) + 1);          // at least 8 MiB per thread
    std::vector<int> rcs;
    try { rcs.assign(nthr, PO_OK); } catch (const std::bad_alloc&) { po_set_error("po_file_read: out of host memory"); return PO_ENOMEM; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { po_set_error("cannot open %s: %s", path, strerror(errno)); return PO_EIO; }
    {   // a freshly allocated destination: ask for huge pages before the reader threads touch it (as in po_api.hip)
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(buf) + 4095u) & ~(uintptr_t)4095u;
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(buf) + len) & ~(uintptr_t)4095u;
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
    const uint64_t per = (len + nthr - 1) / nthr;
    auto work = [&](unsigned t) {
        uint64_t at = (uint64_t)t * per;
        const uint64_t end = std::min<uint64_t>(len, at + per);
        while (at < end) {
            const ssize_t got = pread(fd, buf + at, end - at, (off_t)at);
            if (got <= 0) { rcs[t] = PO_EIO; return; }
            at += (uint64_t)got;
        }
    };
    po_run_threads(nthr, work);
    close(fd);
    for (int rc : rcs)
        if (rc != PO_OK) { po_set_error("read of %s failed: %s", path, strerror(errno)); return rc; }
    return PO_OK;
}

// `rows` pieces of row_bytes bytes, piece r at src + r * src_pitch, to byte offset file_offset + r * file_pitch of an open
// file: the write side of the raw float32 container (bin/phyloligo.py:394-427 assigns row slices of a numpy.memmap, :200-217).
// A row block that spans whole rows of the file (row_bytes == both pitches) is one contiguous range, cut into one piece per
// thread; a block of a tournament work list (columns [c0, c1) of rows [r0, r1), or the transpose of one) is one pwrite per
// row, the rows dealt out in contiguous runs so that every thread walks the file forwards.  pwrite(2) takes its offset as an
// argument: the callers share the descriptor and no file position.
extern "C" int po_pwrite_rows(int fd, const void* src, uint64_t rows, uint64_t row_bytes, uint64_t src_pitch,
                              uint64_t file_offset, uint64_t file_pitch, int threads) {
    if (rows == 0 || row_bytes == 0) return PO_OK;
    if (fd < 0 || !src || src_pitch < row_bytes || file_pitch < row_bytes) {
        po_set_error("po_pwrite_rows: bad argument (fd %d, row_bytes %llu, pitches %llu / %llu)", fd, (unsigned long long)row_bytes,
                     (unsigned long long)src_pitch, (unsigned long long)file_pitch);
        return PO_EINVAL;
    }
    const uint8_t* base = static_cast<const uint8_t*>(src);
    auto put = [&](const uint8_t* p, uint64_t len, uint64_t at) -> int {
        while (len) {                                              // pwrite may write less than asked
            const ssize_t w = pwrite(fd, p, len, (off_t)at);
            if (w < 0) { if (errno == EINTR) continue; return errno ? errno : EIO; }
            if (w == 0) return EIO;                                // no progress and no error: do not spin
            p += w; len -= (uint64_t)w; at += (uint64_t)w;
        }
        return 0;
    };
    const bool contiguous = row_bytes == src_pitch && row_bytes == file_pitch;
    const uint64_t total = rows * row_bytes;
    unsigned nthr = po_host_threads(threads > 0 ? (unsigned)threads : 8u);
    if (total / (4u << 20) + 1 < nthr) nthr = (unsigned)(total / (4u << 20) + 1);      // at least 4 MiB per thread
    if (!contiguous && rows < nthr) nthr = (unsigned)rows;
    std::vector<int> errs;
    try { errs.assign(nthr, 0); } catch (const std::bad_alloc&) { po_set_error("po_pwrite_rows: out of host memory"); return PO_ENOMEM; }
    auto work = [&](unsigned t) {
        if (contiguous) {
            const uint64_t per = ((total + nthr - 1) / nthr + 4095u) & ~(uint64_t)4095u;   // page-aligned cuts
            const uint64_t a = std::min<uint64_t>(total, (uint64_t)t * per), b = std::min<uint64_t>(total, a + per);
            if (b > a) errs[t] = put(base + a, b - a, file_offset + a);
            return;
        }
        const uint64_t per = (rows + nthr - 1) / nthr;
        const uint64_t r0 = std::min<uint64_t>(rows, (uint64_t)t * per), r1 = std::min<uint64_t>(rows, r0 + per);
        for (uint64_t r = r0; r < r1 && !errs[t]; ++r) errs[t] = put(base + r * src_pitch, row_bytes, file_offset + r * file_pitch);
    };
    po_run_threads(nthr, work);
    for (int e : errs)
        if (e) { po_set_error("po_pwrite_rows: pwrite failed: %s", strerror(e)); return PO_EIO; }
    return PO_OK;
}

// The copy threads live for the whole call (spawning 8 threads per 32 MB chunk cost ~20 % of the copy): they wait for
// `ready` to pass their chunk, copy their rows of it and count themselves into `done`; the main thread waits for a
// chunk's producer, for the copies of the chunk before it (whose staging buffer the next fill overwrites), issues the
// next fill and releases the chunk.
int po_ring_copy_rows(const po_ring_source& src, void* const stage[2], size_t stage_bytes, size_t row_bytes, uint64_t rows,
                      uint8_t* dst, size_t dst_pitch, unsigned n_threads) {
    if (rows == 0 || row_bytes == 0) return PO_OK;
    if (row_bytes > stage_bytes || n_threads == 0 || !stage[0] || !stage[1] || !src.issue || !src.wait) {
        po_set_error("po_ring_copy_rows: bad argument");
        return PO_EINVAL;
    }
    const uint64_t rows_per = stage_bytes / row_bytes;
    const uint64_t n_chunks = (rows + rows_per - 1) / rows_per;
    const unsigned n_thr = n_threads;
    auto issue = [&](uint64_t c) -> int {
        const uint64_t r0 = c * rows_per, nr = (rows - r0 < rows_per) ? rows - r0 : rows_per;
        return src.issue(src.user, c, stage[c & 1], r0, nr);
    };
    std::atomic<uint64_t> ready{0}, done{0};
    std::atomic<bool> failed{false};
    std::vector<std::thread> th;
    try {
    th.reserve(n_thr);
    for (unsigned t = 0; t < n_thr; ++t)
        th.emplace_back([&, t]() {
            for (uint64_t c = 0; c < n_chunks; ++c) {
                while (ready.load(std::memory_order_acquire) <= c) {
                    if (failed.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                const uint64_t r0 = c * rows_per, nr = (rows - r0 < rows_per) ? rows - r0 : rows_per;
                const uint8_t* from = static_cast<const uint8_t*>(stage[c & 1]);
                for (uint64_t r = t; r < nr; r += n_thr) memcpy(dst + (r0 + r) * dst_pitch, from + r * row_bytes, row_bytes);
                done.fetch_add(1, std::memory_order_release);
            }
        });
    } catch (...) {                                                   // the system refused a thread: the copies need all of them
        failed.store(true);
        for (auto& x : th) x.join();
        po_set_error("po_ring_copy_rows: could not start %u host threads", n_thr);
        return PO_ENOMEM;
    }
    int rc = issue(0);
    for (uint64_t c = 0; c < n_chunks && rc == PO_OK; ++c) {
        rc = src.wait(src.user);                                      // chunk c is in its staging buffer
        if (rc != PO_OK) break;
        while (done.load(std::memory_order_acquire) < c * n_thr) std::this_thread::yield();   // chunk c - 1 has left its buffer
        if (c + 1 < n_chunks) rc = issue(c + 1);                      // the next fill overlaps the host copies of chunk c
        // a failed issue still releases chunk c: the threads finish it, then see `failed`
        ready.store(c + 1, std::memory_order_release);
    }
    if (rc != PO_OK) failed.store(true);
    for (auto& x : th) x.join();
    return rc;
}

// "%.18e" of one value into p, numpy spelling of non-finite values; returns the new end.
static inline char* format_e18(char* p, double v) {
    if (isnan(v)) { memcpy(p, "nan", 3); return p + 3; }
    if (isinf(v)) {
        if (v < 0) { memcpy(p, "-inf", 4); return p + 4; }
        memcpy(p, "inf", 3);
        return p + 3;
    }
    // std::to_chars(scientific, 18) prints exactly what printf("%.18e") prints (correctly rounded digits,
    // at least two exponent digits) at a fraction of the cost
    auto r = std::to_chars(p, p + 32, v, std::chars_format::scientific, 18);
    return r.ptr;
}

// Formats rows [r0, r1) into `out` (cleared first): straight into the buffer, 27 bytes at most per entry ("-d.<18 digits>e-308" = 26
// + separator; format_e18 may look 32 bytes ahead).
static void format_rows(const double* m, uint64_t r0, uint64_t r1, uint64_t cols, uint64_t ld, std::vector<char>& out) {
    out.resize((size_t)((r1 - r0) * cols * 27 + 40));
    char* p = out.data();
    for (uint64_t r = r0; r < r1; ++r) {
        const double* row = m + r * ld;
        for (uint64_t c = 0; c + 1 < cols; ++c) {
            p = format_e18(p, row[c]);
            *p++ = '\t';
        }
        p = format_e18(p, row[cols - 1]);
        *p++ = '\n';
    }
    out.resize((size_t)(p - out.data()));
}

// numpy.savetxt(path, m, delimiter="\t") (bin/phyloligo.py:1059-1066), byte for byte.  The rows are formatted by the host threads the
// job may use, a slab of rows each; the sizes of a round's slabs give their places in the file and the same threads pwrite them
// there (round 4: the slabs used to go through ONE fwrite in order - 1.1 GB/s of text on the gpurun box, 22.8 ns per entry; a
// 12 000 x 12 000 matrix took 3.3 s).  A file that exists is overwritten in place and cut to the new length at the end (O_TRUNC
// on a large cached file costs as much as writing it, see phyloligo.py:_open_raw_container); append = 1 continues at its end.
extern "C" int po_write_mat_text(const double* m, uint64_t rows, uint64_t cols, uint64_t ld, const char* path,
                                 int append) {
    if (!path || (!m && rows && cols) || ld < cols) {
        po_set_error("po_write_mat_text: bad argument");
        return PO_EINVAL;
    }
    const int fd = open(path, O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
    if (fd < 0) {
        po_set_error("cannot open %s: %s", path, strerror(errno));
        return PO_EIO;
    }
    // A regular file takes its slabs by pwrite at their places; anything else (/dev/stdout into a pipe, a FIFO, /dev/null) gets
    // them in order through write() from one thread, as numpy.savetxt would.
    struct stat sb;
    const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    uint64_t at = 0;                                               // where the next byte goes
    if (append && regular) {
        const off_t end = lseek(fd, 0, SEEK_END);
        if (end < 0) { po_set_error("cannot seek in %s: %s", path, strerror(errno)); close(fd); return PO_EIO; }
        at = (uint64_t)end;
    }
    int rc = PO_OK, io_errno = 0;
    if (rows && cols) {
        // Slabs of rows (<= 8 MB of text) are taken in order by the threads; a slab's place in the file is the end of the one
        // before it, known as soon as THAT one is formatted - so a thread formats, waits for its place (the chain runs through
        // slabs taken earlier, by threads that wait for nothing later), hands the next place on and pwrites, while the others
        // are formatting: no round structure, no serial write.
        const unsigned nthreads = (rows * cols < (1u << 16) || !regular) ? 1u : po_host_threads(32);
        const uint64_t slab = std::max<uint64_t>(1, std::min<uint64_t>((rows + nthreads - 1) / nthreads, (8u << 20) / (cols * 25 + 1) + 1));
        const uint64_t n_slabs = (rows + slab - 1) / slab;
        constexpr uint64_t NOT_YET = ~0ull;
        std::unique_ptr<std::atomic<uint64_t>[]> place;
        std::vector<std::vector<char>> bufs;
        try {
            place.reset(new std::atomic<uint64_t>[n_slabs + 1]);
            bufs.resize(nthreads);
        } catch (const std::bad_alloc&) { rc = PO_ENOMEM; }
        if (rc == PO_OK) {
            for (uint64_t i = 0; i <= n_slabs; ++i) place[i].store(i == 0 ? at : NOT_YET, std::memory_order_relaxed);
            std::atomic<uint64_t> next{0};
            std::atomic<int> failed{0};                                 // 0, or -1 (out of memory), or the errno of a write
            po_run_threads(nthreads, [&](unsigned t) {
                for (;;) {
                    const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
                    if (i >= n_slabs || failed.load(std::memory_order_relaxed)) return;
                    const uint64_t r0 = i * slab, r1 = std::min(rows, r0 + slab);
                    try { format_rows(m, r0, r1, cols, ld, bufs[t]); } catch (...) { failed.store(-1); return; }
                    uint64_t off;
                    while ((off = place[i].load(std::memory_order_acquire)) == NOT_YET) {
                        if (failed.load(std::memory_order_relaxed)) return;
                        std::this_thread::yield();
                    }
                    place[i + 1].store(off + bufs[t].size(), std::memory_order_release);
                    const char* p = bufs[t].data();
                    uint64_t len = bufs[t].size();
                    while (len) {
                        const ssize_t w = regular ? pwrite(fd, p, len, (off_t)off) : write(fd, p, len);
                        if (w < 0 && errno == EINTR) continue;
                        if (w <= 0) { failed.store(w < 0 && errno ? errno : EIO); return; }
                        p += w; len -= (uint64_t)w; off += (uint64_t)w;
                    }
                }
            });
            const int f = failed.load();
            if (f == -1) rc = PO_ENOMEM;
            else if (f) { rc = PO_EIO; io_errno = f; }
            else at = place[n_slabs].load();
        }
    }
    if (rc == PO_OK && regular && ftruncate(fd, (off_t)at) != 0) { rc = PO_EIO; io_errno = errno; }     // an older, longer file ends here
    if (close(fd) != 0 && rc == PO_OK) { rc = PO_EIO; io_errno = errno; }
    if (rc == PO_ENOMEM) po_set_error("po_write_mat_text: out of host memory while formatting %s", path);
    else if (rc != PO_OK) po_set_error("write to %s failed: %s", path, strerror(io_errno));
    return rc;
}
