// Sanitizer harness for the host-only code of libphyloligo_amd.so (po_io.cpp): the segment-parallel FASTA parser, the
// threaded .mat writer, the parallel file reader and the two-buffer copy ring of the device-to-host path.  Built by
// `make san SAN=address,undefined` / `make san SAN=thread` with clang++ (no HIP), driven by tools/run_sanitizers.py.
//
//   san_host_test fasta <file>        parse with po_fasta_scan / po_fasta_extract; prints records, sequence bytes, an
//                                     FNV-1a hash of (titles, offsets, sequence) and checks the result against a plain
//                                     sequential restatement of the same rules in this file
//   san_host_test bigfasta <MiB>      the same on a generated multi-segment file of that size (CRLF / blank / wrapped mix)
//   san_host_test mat <rows> <cols> <path>   po_write_mat_text of a generated matrix, read back and compared with snprintf("%.18e")
//   san_host_test fileread <path>     po_file_read against a plain fread
//   san_host_test ring <MiB>          po_ring_copy_rows with a producer thread filling the staging buffers
//   san_host_test pwrite <n> <path>   po_pwrite_rows: an n x n float32 container written as whole-row blocks and as
//                                     rectangular blocks (+ transposes), read back and compared entry by entry
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "phyloligo_amd.h"
#include "../po_host.h"

static thread_local char g_err[512];
void po_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "CHECK failed: %s (%s:%d) %s\n", #cond, __FILE__, __LINE__, g_err); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static bool py_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

// one sequential walk: the rules of csrc/po_io.cpp's header comment, written independently of its segment logic
static int plain_parse(const std::vector<uint8_t>& d, std::vector<std::string>& titles, std::vector<uint64_t>& off, std::string& seq) {
    size_t pos = 0;
    bool in_rec = false;
    while (pos < d.size()) {
        size_t e = pos;
        while (e < d.size() && d[e] != '\n') ++e;
        size_t r = e;
        while (r > pos && py_space(d[r - 1])) --r;
        if (d[pos] == '>' && e > pos) {
            off.push_back(seq.size());
            titles.emplace_back(reinterpret_cast<const char*>(d.data()) + pos + 1, r > pos + 1 ? r - pos - 1 : 0);
            in_rec = true;
        } else if (!in_rec) {
            if (r > pos) return -1;
        } else {
            for (size_t i = pos; i < r; ++i)
                if (d[i] != ' ' && d[i] != '\r') seq.push_back((char)d[i]);
        }
        pos = e < d.size() ? e + 1 : e;
    }
    off.push_back(seq.size());
    return 0;
}

static int run_fasta(const std::vector<uint8_t>& d) {
    uint64_t nrec = 0, nbytes = 0;
    std::vector<std::string> titles;
    std::vector<uint64_t> want_off;
    std::string want_seq;
    const int want_rc = plain_parse(d, titles, want_off, want_seq);
    int rc = po_fasta_scan(d.data(), d.size(), &nrec, &nbytes);
    if (want_rc) { CHECK(rc == PO_EIO); printf("rejected (text before the first record)\n"); return 0; }
    CHECK(rc == PO_OK);
    CHECK(nrec == titles.size() && nbytes == want_seq.size());
    std::vector<uint8_t> seq(nbytes ? nbytes : 1);
    std::vector<uint64_t> off(nrec + 1), tb(nrec ? nrec : 1), te(nrec ? nrec : 1);
    rc = po_fasta_extract(d.data(), d.size(), seq.data(), off.data(), tb.data(), te.data());
    CHECK(rc == PO_OK);
    CHECK(memcmp(off.data(), want_off.data(), (nrec + 1) * 8) == 0);
    CHECK(nbytes == 0 || memcmp(seq.data(), want_seq.data(), nbytes) == 0);
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < nrec; ++i) {
        CHECK(te[i] >= tb[i] && te[i] <= d.size());
        CHECK(te[i] - tb[i] == titles[i].size() && memcmp(d.data() + tb[i], titles[i].data(), titles[i].size()) == 0);
        h = fnv(h, titles[i].data(), titles[i].size());
    }
    h = fnv(h, off.data(), (nrec + 1) * 8);
    h = fnv(h, seq.data(), nbytes);
    printf("records %llu seq_bytes %llu hash %016llx\n", (unsigned long long)nrec, (unsigned long long)nbytes, (unsigned long long)h);
    return 0;
}

static uint64_t g_rng = 88172645463325252ull;
static uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 16); }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: see the header of san_host_test.cpp\n"); return 2; }
    const std::string mode = argv[1];
    if (mode == "fasta" && argc == 3) {
        FILE* fh = fopen(argv[2], "rb");
        CHECK(fh != nullptr);
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), fh)) > 0) d.insert(d.end(), buf, buf + got);
        fclose(fh);
        return run_fasta(d);
    }
    if (mode == "bigfasta" && argc == 3) {
        const size_t target = (size_t)atol(argv[2]) << 20;
        std::vector<uint8_t> d;
        d.reserve(target + 4096);
        const char* alpha = "ACGTNacgtn";
        for (const char* c = "\n  \n\r\n"; *c; ++c) d.push_back((uint8_t)*c);      // blank prelude
        for (unsigned rec = 0; d.size() < target; ++rec) {
            char t[96];
            const int tl = snprintf(t, sizeof(t), ">rec%u some text %u  ", rec, rnd() % 1000);
            d.insert(d.end(), t, t + tl);
            const bool crlf = rec % 5 == 0;
            if (crlf) d.push_back('\r');
            d.push_back('\n');
            const uint32_t len = (rec % 11 == 0) ? 0 : (rec % 7 == 0 ? 300000 + rnd() % 100000 : rnd() % 6000);
            const uint32_t width = rec % 3 == 0 ? 0xFFFFFFFFu : 1 + rnd() % 120;
            for (uint32_t i = 0; i < len; ++i) {
                d.push_back((uint8_t)alpha[rnd() % 10]);
                if (rnd() % 400 == 0) d.push_back(' ');
                if ((i + 1) % width == 0 || i + 1 == len) { if (crlf) d.push_back('\r'); d.push_back('\n'); }
            }
            if (rec % 13 == 0) d.push_back('\n');
        }
        if (rnd() & 1) d.pop_back();                                               // sometimes no final newline
        printf("generated %zu bytes\n", d.size());
        return run_fasta(d);
    }
    if (mode == "mat" && argc == 5) {
        const uint64_t rows = strtoull(argv[2], nullptr, 10), cols = strtoull(argv[3], nullptr, 10);
        std::vector<double> m(rows * cols);
        for (auto& v : m) {
            const uint32_t r = rnd();
            v = (r % 97 == 0) ? 0.0 : (r % 1013 == 0 ? -(double)r : (double)r / 4294967296.0 * (r % 5 == 0 ? 1e-7 : 1.0));
        }
        if (m.size() > 8) { m[3] = NAN; m[5] = INFINITY; m[6] = -INFINITY; m[7] = 5e-324; }
        CHECK(po_write_mat_text(m.data(), rows, cols, cols, argv[4], 0) == PO_OK);
        FILE* fh = fopen(argv[4], "rb");
        CHECK(fh != nullptr);
        std::vector<char> line(cols * 32 + 16);
        uint64_t checked = 0;
        for (uint64_t r = 0; r < rows; ++r) {
            CHECK(fgets(line.data(), (int)line.size(), fh) != nullptr);
            if (r % 97 && r + 1 != rows && r > 2) continue;                        // compare a sample of rows with printf
            std::string want;
            for (uint64_t c = 0; c < cols; ++c) {
                char t[48];
                const double v = m[r * cols + c];
                if (v != v) snprintf(t, sizeof(t), "nan");
                else if (v == INFINITY) snprintf(t, sizeof(t), "inf");
                else if (v == -INFINITY) snprintf(t, sizeof(t), "-inf");
                else snprintf(t, sizeof(t), "%.18e", v);
                want += t;
                want += (c + 1 == cols) ? '\n' : '\t';
            }
            CHECK(want == line.data());
            ++checked;
        }
        CHECK(fgetc(fh) == EOF);
        fclose(fh);
        printf("wrote %llu x %llu, %llu rows compared with printf\n", (unsigned long long)rows, (unsigned long long)cols, (unsigned long long)checked);
        return 0;
    }
    if (mode == "fileread" && argc == 3) {
        FILE* fh = fopen(argv[2], "rb");
        CHECK(fh != nullptr);
        std::vector<uint8_t> want;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), fh)) > 0) want.insert(want.end(), buf, buf + got);
        fclose(fh);
        std::vector<uint8_t> have(want.size() ? want.size() : 1);
        CHECK(po_file_read(argv[2], have.data(), want.size()) == PO_OK);
        CHECK(want.empty() || memcmp(have.data(), want.data(), want.size()) == 0);
        CHECK(po_file_read("/nonexistent/file", have.data(), 1) == PO_EIO);
        printf("read %zu bytes\n", want.size());
        return 0;
    }
    if (mode == "pwrite" && argc == 4) {
        const uint64_t n = strtoull(argv[2], nullptr, 10);
        CHECK(n >= 8);
        std::vector<float> m(n * n);
        for (uint64_t i = 0; i < n * n; ++i) m[i] = (float)(i % 1000003) * 0.25f;
        FILE* mk = fopen(argv[3], "wb");
        CHECK(mk != nullptr);
        fclose(mk);
        const int fd = open(argv[3], O_RDWR);
        CHECK(fd >= 0);
        CHECK(ftruncate(fd, (off_t)(n * n * 4)) == 0);
        const uint64_t h = n / 2, q = n / 3;
        // rows [0, h): whole rows, one contiguous range, 5 threads; from a source with the same pitch
        CHECK(po_pwrite_rows(fd, m.data(), h, n * 4, n * 4, 0, n * 4, 5) == PO_OK);
        // rows [h, n) x columns [0, q): a rectangular block out of a packed [n - h, q] buffer (what a mirror block is)
        std::vector<float> blk((n - h) * q);
        for (uint64_t r = h; r < n; ++r) memcpy(&blk[(r - h) * q], &m[r * n], q * 4);
        CHECK(po_pwrite_rows(fd, blk.data(), n - h, q * 4, q * 4, h * n * 4, n * 4, 7) == PO_OK);
        // rows [h, n) x columns [q, n): straight out of the matrix (source pitch = n floats), default thread count
        CHECK(po_pwrite_rows(fd, &m[h * n + q], n - h, (n - q) * 4, n * 4, (h * n + q) * 4, n * 4, 0) == PO_OK);
        CHECK(po_pwrite_rows(fd, m.data(), 0, 16, 16, 0, 16, 4) == PO_OK);                 // nothing to do
        CHECK(po_pwrite_rows(-1, m.data(), 1, 16, 16, 0, 16, 4) == PO_EINVAL);
        CHECK(po_pwrite_rows(fd, m.data(), 2, 16, 8, 0, 16, 4) == PO_EINVAL);              // pitch below the row
        close(fd);
        const int ro = open(argv[3], O_RDONLY);
        CHECK(ro >= 0);
        CHECK(po_pwrite_rows(ro, m.data(), 1, 16, 16, 0, 16, 1) == PO_EIO);                // not writable
        close(ro);
        std::vector<float> back(n * n);
        CHECK(po_file_read(argv[3], reinterpret_cast<uint8_t*>(back.data()), n * n * 4) == PO_OK);
        CHECK(memcmp(back.data(), m.data(), n * n * 4) == 0);
        printf("pwrite %llu x %llu float32 ok\n", (unsigned long long)n, (unsigned long long)n);
        return 0;
    }
    if (mode == "ring" && argc == 3) {
        // the "device": a matrix with a pitch; the producer is a thread that fills the staging buffer some time after issue()
        const size_t row_bytes = 40000 * 4, pitch = row_bytes + 256;
        const uint64_t rows = ((size_t)atol(argv[2]) << 20) / row_bytes + 3;
        std::vector<uint8_t> dev(rows * pitch), dst(rows * (row_bytes + 64), 0xEE);
        for (size_t i = 0; i < dev.size(); ++i) dev[i] = (uint8_t)(i * 2654435761u >> 13);
        const size_t stage_bytes = 8u << 20;
        std::vector<uint8_t> s0(stage_bytes), s1(stage_bytes);
        void* stage[2] = {s0.data(), s1.data()};
        struct prod { const uint8_t* dev; size_t pitch, row_bytes; std::thread th; } u{dev.data(), pitch, row_bytes, {}};
        po_ring_source src;
        src.user = &u;
        src.issue = [](void* p, uint64_t, void* st, uint64_t r0, uint64_t nr) -> int {
            prod* q = static_cast<prod*>(p);
            q->th = std::thread([q, st, r0, nr]() {
                for (uint64_t r = 0; r < nr; ++r) memcpy(static_cast<uint8_t*>(st) + r * q->row_bytes, q->dev + (r0 + r) * q->pitch, q->row_bytes);
            });
            return PO_OK;
        };
        src.wait = [](void* p) -> int { static_cast<prod*>(p)->th.join(); return PO_OK; };
        for (unsigned n_thr : {1u, 3u, 14u}) {
            std::fill(dst.begin(), dst.end(), (uint8_t)0xEE);
            CHECK(po_ring_copy_rows(src, stage, stage_bytes, row_bytes, rows, dst.data(), row_bytes + 64, n_thr) == PO_OK);
            for (uint64_t r = 0; r < rows; ++r) {
                CHECK(memcmp(dst.data() + r * (row_bytes + 64), dev.data() + r * pitch, row_bytes) == 0);
                CHECK(dst[r * (row_bytes + 64) + row_bytes] == 0xEE);              // nothing written past a row
            }
        }
        // a producer that fails on its third chunk: the ring stops, the threads leave, the error comes back
        struct bad { int calls; } b{0};
        po_ring_source fs;
        fs.user = &b;
        fs.issue = [](void* p, uint64_t, void*, uint64_t, uint64_t) -> int { return ++static_cast<bad*>(p)->calls >= 3 ? PO_EHIP : PO_OK; };
        fs.wait = [](void*) -> int { return PO_OK; };
        CHECK(po_ring_copy_rows(fs, stage, stage_bytes, row_bytes, rows, dst.data(), row_bytes + 64, 6) == PO_EHIP);
        printf("ring copied %llu rows x %zu bytes with 1, 3 and 14 threads; failing producer handled\n", (unsigned long long)rows, row_bytes);
        return 0;
    }
    fprintf(stderr, "unknown mode\n");
    return 2;
}
