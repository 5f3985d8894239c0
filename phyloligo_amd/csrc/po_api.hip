// C ABI of libphyloligo_amd.so (include/phyloligo_amd.h): contexts, argument checking, the
// host-pointer convenience forms, and the dispatch of a pairwise request onto the tile kernels.
#include "po_internal.h"
#include "po_host.h"

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include <stdarg.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <new>

// ---- errors --------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void po_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* po_last_error(void) { return g_err; }

extern "C" const char* po_status_string(int status) {
    switch (status) {
        case PO_OK: return "ok";
        case PO_EINVAL: return "invalid argument";
        case PO_ENODEV: return "no usable HIP device";
        case PO_ENOMEM: return "out of memory";
        case PO_EHIP: return "HIP runtime error";
        case PO_EUNSUPPORTED: return "unsupported request";
        case PO_EIO: return "I/O error";
        default: return "unknown status";
    }
}

extern "C" const char* po_version(void) { return "phyloligo_amd 0.1 (gfx950)"; }
extern "C" int po_abi_version(void) { return PO_ABI_VERSION; }

extern "C" int po_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- context -------------------------------------------------------------------------------
extern "C" int po_ctx_create(po_ctx** out, int device_id) {
    PO_REQUIRE(out != nullptr, "po_ctx_create: out is NULL");
    *out = nullptr;
    const int n = po_device_count();
    if (n <= 0) {
        po_set_error("po_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return PO_ENODEV;
    }
    PO_REQUIRE(device_id >= 0 && device_id < n, "po_ctx_create: device %d out of range (0..%d)", device_id, n - 1);
    po_ctx* c = new (std::nothrow) po_ctx();
    if (!c) { po_set_error("po_ctx_create: host allocation failed"); return PO_ENOMEM; }
    c->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        po_set_error("po_ctx_create: %s", hipGetErrorString(e));
        delete c;
        return PO_EHIP;
    }
    *out = c;
    return PO_OK;
}

static void buf_free(po_buf* b) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
}

extern "C" void po_ctx_destroy(po_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    buf_free(&ctx->ws_freq);
    buf_free(&ctx->ws_rowstat);
    buf_free(&ctx->ws_aux);
    buf_free(&ctx->ws_io);
    buf_free(&ctx->ws_logtab);
    buf_free(&ctx->ws_fold);
    buf_free(&ctx->ws_fold_src);
    buf_free(&ctx->ws_recover);
    buf_free(&ctx->ws_pairdot);
    buf_free(&ctx->ws_pq);
    buf_free(&ctx->ws_thermo);
    buf_free(&ctx->ws_fasta);
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    for (int i = 0; i < 2; ++i)
        if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
    for (int i = 0; i < 4; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
}

extern "C" int po_ctx_trim(po_ctx* ctx) {
    PO_REQUIRE(ctx != nullptr, "po_ctx_trim: ctx is NULL");
    PO_HIP(hipSetDevice(ctx->device));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    po_buf* all[] = {&ctx->ws_freq, &ctx->ws_rowstat, &ctx->ws_aux, &ctx->ws_io, &ctx->ws_fold, &ctx->ws_fold_src, &ctx->ws_recover,
                     &ctx->ws_pairdot, &ctx->ws_pq, &ctx->ws_thermo, &ctx->ws_fasta};
    for (po_buf* b : all) buf_free(b);
    buf_free(&ctx->ws_logtab);              // rebuilt by po_logtab_init on the next JSD call
    ctx->logtab_ready = false;
    for (int i = 0; i < 2; ++i)             // the pinned ring of the host-pointer forms (2 x 32 MB), re-created on demand
        if (ctx->h_stage[i]) { (void)hipHostFree(ctx->h_stage[i]); ctx->h_stage[i] = nullptr; }
    ctx->pq_key = ~0ull;                    // cached tables went with their buffers
    ctx->fold_dim = ctx->fold_gran = ctx->fold_dim_f = ctx->fold_dbl_at = 0;
    ctx->fasta_data = nullptr;
    ctx->fasta_len = ctx->fasta_records = ctx->fasta_seq_bytes = 0;
    return PO_OK;
}

extern "C" int po_ctx_set_stream(po_ctx* ctx, void* hip_stream) {
    PO_REQUIRE(ctx != nullptr, "po_ctx_set_stream: ctx is NULL");
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return PO_OK;
}

extern "C" int po_ctx_synchronize(po_ctx* ctx) {
    PO_REQUIRE(ctx != nullptr, "po_ctx_synchronize: ctx is NULL");
    PO_HIP(hipSetDevice(ctx->device));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    return PO_OK;
}

extern "C" int po_ctx_device_name(po_ctx* ctx, char* buf, size_t len) {
    PO_REQUIRE(ctx != nullptr && buf != nullptr && len > 0, "po_ctx_device_name: bad argument");
    snprintf(buf, len, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
    return PO_OK;
}

int po_buf_reserve(po_ctx* ctx, po_buf* b, size_t bytes) {
    if (bytes <= b->cap) return PO_OK;
    // growing is rare (first call for a problem size): finish queued work that may still use the old block
    PO_HIP(hipStreamSynchronize(ctx->stream));
    buf_free(b);
    const size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
        b->p = nullptr;
        (void)hipGetLastError();               // the failed allocation must not surface again at the next launch check
        po_set_error("device allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return PO_ENOMEM;
    }
    b->cap = want;
    return PO_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel function of the process (per device), not to a
// context: two contexts on one device must see one table, or the second one could lower a limit the first relies on.
// The granted size only ever grows.
namespace {
struct shmem_entry { int device; const void* func; size_t bytes; };
std::mutex g_shmem_mu;
std::vector<shmem_entry> g_shmem;
}

int po_func_shmem(po_ctx* ctx, const void* func, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_shmem_mu);
    for (auto& e : g_shmem)
        if (e.device == ctx->device && e.func == func) {
            if (e.bytes >= bytes) return PO_OK;
            PO_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.bytes = bytes;
            return PO_OK;
        }
    PO_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    g_shmem.push_back({ctx->device, func, bytes});
    return PO_OK;
}

// ---- pattern -------------------------------------------------------------------------------
int po_pattern_compile(const char* pattern, po_pattern* out) {
    PO_REQUIRE(pattern != nullptr && out != nullptr, "pattern is NULL");
    memset(out, 0, sizeof(*out));
    const size_t W = strlen(pattern);
    PO_REQUIRE(W >= 1, "pattern is empty");
    for (size_t i = 0; i < W; ++i)
        PO_REQUIRE(pattern[i] == '0' || pattern[i] == '1', "pattern '%s' must contain only 0 and 1", pattern);
    if (W > PO_MAX_WINDOW) { po_set_error("pattern longer than %d positions is not supported", PO_MAX_WINDOW); return PO_EUNSUPPORTED; }
    uint32_t k = 0;
    for (size_t i = 0; i < W; ++i)
        if (pattern[i] == '1') out->ones[k++] = (uint32_t)i;     // k <= W <= PO_MAX_WINDOW
    PO_REQUIRE(k >= 1, "pattern '%s' selects no position", pattern);
    if (k > PO_MAX_K) { po_set_error("patterns with more than %d selected positions (4^k words) are not supported", PO_MAX_K); return PO_EUNSUPPORTED; }
    out->window = (uint32_t)W;
    out->k = k;
    out->dim = 1u << (2 * k);
    // runs of consecutive '1': window position x sits at bits [2(W-1-x), +2) of the rolling register
    uint32_t rank = 0, nruns = 0;
    for (size_t x = 0; x < W;) {
        if (pattern[x] != '1') { ++x; continue; }
        size_t y = x;
        while (y < W && pattern[y] == '1') ++y;
        const uint32_t len = (uint32_t)(y - x);
        out->src_shift[nruns] = 2 * (uint32_t)(W - x - len);
        out->dst_shift[nruns] = 2 * (k - rank - len);
        out->mask[nruns] = (len >= 16) ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u);
        ++nruns;
        rank += len;
        x = y;
    }
    out->nruns = nruns;
    return PO_OK;
}

extern "C" int po_pattern_info(const char* pattern, uint32_t* window, uint32_t* k, uint64_t* dim) {
    po_pattern p;
    int rc = po_pattern_compile(pattern, &p);
    if (rc) return rc;
    if (window) *window = p.window;
    if (k) *k = p.k;
    if (dim) *dim = p.dim;
    return PO_OK;
}

static int check_strand(int strand) {
    // select_strand prints an error and exits(1) for anything else (phyloligo.py:146-148)
    PO_REQUIRE(strand == PO_STRAND_BOTH || strand == PO_STRAND_PLUS || strand == PO_STRAND_MINUS,
               "strand must be one of both/plus/minus (got %d)", strand);
    return PO_OK;
}

static int check_metric(int metric) {
    // compute_distances_joblib prints "unknown metric" and exits(1) (phyloligo.py:383-385)
    PO_REQUIRE(metric >= PO_EUCL && metric <= PO_SC, "unknown metric %d (Eucl=0, JSD=1, KT=2, BC=3, SC=4)", metric);
    return PO_OK;
}

// ---- stage 1 -------------------------------------------------------------------------------
extern "C" int po_count_profiles_dev(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t n_seqs,
                                     uint64_t total_bytes, const char* pattern, int strand, uint32_t* d_counts,
                                     uint64_t* d_totals) {
    PO_REQUIRE(ctx != nullptr, "po_count_profiles_dev: ctx is NULL");
    po_pattern pat;
    int rc = po_pattern_compile(pattern, &pat);
    if (rc) return rc;
    rc = check_strand(strand);
    if (rc) return rc;
    if (n_seqs == 0) return PO_OK;
    PO_REQUIRE(d_offsets && d_counts && d_totals, "po_count_profiles_dev: NULL buffer");
    PO_REQUIRE(d_seq != nullptr || total_bytes == 0, "po_count_profiles_dev: NULL sequence buffer");
    PO_REQUIRE((reinterpret_cast<uintptr_t>(d_seq) & 15u) == 0, "po_count_profiles_dev: sequence buffer must be 16-byte aligned");
    PO_HIP(hipSetDevice(ctx->device));
    return po_launch_count(ctx, d_seq, d_offsets, d_offsets + 1, n_seqs, total_bytes, total_bytes, pat, strand, d_counts, d_totals);
}

extern "C" int po_count_profiles(po_ctx* ctx, const uint8_t* seq, const uint64_t* offsets, uint64_t n_seqs,
                                 const char* pattern, int strand, uint32_t* counts, uint64_t* totals) {
    PO_REQUIRE(ctx != nullptr, "po_count_profiles: ctx is NULL");
    po_pattern pat;
    int rc = po_pattern_compile(pattern, &pat);
    if (rc) return rc;
    rc = check_strand(strand);
    if (rc) return rc;
    if (n_seqs == 0) return PO_OK;
    PO_REQUIRE(offsets && counts && totals, "po_count_profiles: NULL buffer");
    PO_REQUIRE(offsets[0] == 0, "po_count_profiles: offsets[0] must be 0");
    for (uint64_t i = 0; i < n_seqs; ++i)
        PO_REQUIRE(offsets[i + 1] >= offsets[i], "po_count_profiles: offsets must be non-decreasing (record %llu)", (unsigned long long)i);
    const uint64_t total = offsets[n_seqs];
    PO_REQUIRE(seq != nullptr || total == 0, "po_count_profiles: NULL sequence buffer");
    PO_HIP(hipSetDevice(ctx->device));

    const size_t b_seq = po_round_up(total + 64, 256);
    const size_t b_off = po_round_up((n_seqs + 1) * sizeof(uint64_t), 256);
    const size_t b_cnt = po_round_up(n_seqs * (uint64_t)pat.dim * sizeof(uint32_t), 256);
    const size_t b_tot = po_round_up(n_seqs * sizeof(uint64_t), 256);
    rc = po_buf_reserve(ctx, &ctx->ws_io, b_seq + b_off + b_cnt + b_tot);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_io.p);
    uint8_t* d_seq = base;
    uint64_t* d_off = reinterpret_cast<uint64_t*>(base + b_seq);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base + b_seq + b_off);
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(base + b_seq + b_off + b_cnt);
    if (total) PO_HIP(hipMemcpyAsync(d_seq, seq, total, hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_off, offsets, (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = po_launch_count(ctx, d_seq, d_off, d_off + 1, n_seqs, total, total, pat, strand, d_cnt, d_tot);
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(counts, d_cnt, n_seqs * (uint64_t)pat.dim * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipMemcpyAsync(totals, d_tot, n_seqs * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    return PO_OK;
}

extern "C" int po_frequencies_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                                  uint32_t dim, double* d_freq) {
    PO_REQUIRE(ctx != nullptr, "po_frequencies_dev: ctx is NULL");
    if (n == 0) return PO_OK;
    PO_REQUIRE(d_counts && d_totals && d_freq && dim > 0, "po_frequencies_dev: bad argument");
    PO_HIP(hipSetDevice(ctx->device));
    return po_launch_freq_rowmajor(ctx, d_counts, d_totals, n, dim, d_freq);
}

extern "C" int po_frequencies(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                              double* freq) {
    PO_REQUIRE(ctx != nullptr, "po_frequencies: ctx is NULL");
    if (n == 0) return PO_OK;
    PO_REQUIRE(counts && totals && freq && dim > 0, "po_frequencies: bad argument");
    PO_HIP(hipSetDevice(ctx->device));
    const size_t b_cnt = po_round_up(n * (uint64_t)dim * sizeof(uint32_t), 256);
    const size_t b_tot = po_round_up(n * sizeof(uint64_t), 256);
    const size_t b_frq = n * (uint64_t)dim * sizeof(double);
    int rc = po_buf_reserve(ctx, &ctx->ws_io, b_cnt + b_tot + b_frq);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_io.p);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base);
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(base + b_cnt);
    double* d_frq = reinterpret_cast<double*>(base + b_cnt + b_tot);
    PO_HIP(hipMemcpyAsync(d_cnt, counts, n * (uint64_t)dim * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_tot, totals, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = po_launch_freq_rowmajor(ctx, d_cnt, d_tot, n, dim, d_frq);
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(freq, d_frq, b_frq, hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    return PO_OK;
}

// ---- windows (Kount.py) -------------------------------------------------------------------------
extern "C" int po_count_profiles_ranges_dev(po_ctx* ctx, const uint8_t* d_seq, uint64_t total_bytes,
                                            const uint64_t* d_begins, const uint64_t* d_ends, uint64_t n_ranges,
                                            uint64_t sum_lengths, const char* pattern, int strand,
                                            uint32_t* d_counts, uint64_t* d_totals) {
    PO_REQUIRE(ctx != nullptr, "po_count_profiles_ranges_dev: ctx is NULL");
    po_pattern pat;
    int rc = po_pattern_compile(pattern, &pat);
    if (rc) return rc;
    rc = check_strand(strand);
    if (rc) return rc;
    if (n_ranges == 0) return PO_OK;
    PO_REQUIRE(d_begins && d_ends && d_counts && d_totals, "po_count_profiles_ranges_dev: NULL buffer");
    PO_REQUIRE(d_seq != nullptr || total_bytes == 0, "po_count_profiles_ranges_dev: NULL sequence buffer");
    PO_REQUIRE((reinterpret_cast<uintptr_t>(d_seq) & 15u) == 0, "po_count_profiles_ranges_dev: sequence buffer must be 16-byte aligned");
    PO_HIP(hipSetDevice(ctx->device));
    return po_launch_count(ctx, d_seq, d_begins, d_ends, n_ranges, total_bytes, sum_lengths, pat, strand, d_counts, d_totals);
}

extern "C" int po_count_profiles_ranges(po_ctx* ctx, const uint8_t* seq, uint64_t total_bytes, const uint64_t* begins,
                                        const uint64_t* ends, uint64_t n_ranges, const char* pattern, int strand,
                                        uint32_t* counts, uint64_t* totals) {
    PO_REQUIRE(ctx != nullptr, "po_count_profiles_ranges: ctx is NULL");
    po_pattern pat;
    int rc = po_pattern_compile(pattern, &pat);
    if (rc) return rc;
    rc = check_strand(strand);
    if (rc) return rc;
    if (n_ranges == 0) return PO_OK;
    PO_REQUIRE(begins && ends && counts && totals, "po_count_profiles_ranges: NULL buffer");
    PO_REQUIRE(seq != nullptr || total_bytes == 0, "po_count_profiles_ranges: NULL sequence buffer");
    uint64_t sum = 0;
    for (uint64_t i = 0; i < n_ranges; ++i) {
        PO_REQUIRE(begins[i] <= ends[i] && ends[i] <= total_bytes, "po_count_profiles_ranges: range %llu outside the buffer", (unsigned long long)i);
        sum += ends[i] - begins[i];
    }
    PO_HIP(hipSetDevice(ctx->device));
    const size_t b_seq = po_round_up(total_bytes + 64, 256);
    const size_t b_rng = po_round_up(n_ranges * sizeof(uint64_t), 256);
    const size_t b_cnt = po_round_up(n_ranges * (uint64_t)pat.dim * sizeof(uint32_t), 256);
    const size_t b_tot = po_round_up(n_ranges * sizeof(uint64_t), 256);
    rc = po_buf_reserve(ctx, &ctx->ws_io, b_seq + 2 * b_rng + b_cnt + b_tot);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_io.p);
    uint8_t* d_seq = base;
    uint64_t* d_beg = reinterpret_cast<uint64_t*>(base + b_seq);
    uint64_t* d_end = reinterpret_cast<uint64_t*>(base + b_seq + b_rng);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base + b_seq + 2 * b_rng);
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(base + b_seq + 2 * b_rng + b_cnt);
    if (total_bytes) PO_HIP(hipMemcpyAsync(d_seq, seq, total_bytes, hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_beg, begins, n_ranges * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_end, ends, n_ranges * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = po_launch_count(ctx, d_seq, d_beg, d_end, n_ranges, total_bytes, sum, pat, strand, d_cnt, d_tot);
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(counts, d_cnt, n_ranges * (uint64_t)pat.dim * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipMemcpyAsync(totals, d_tot, n_ranges * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    return PO_OK;
}

extern "C" int po_count_byte_ranges_dev(po_ctx* ctx, const uint8_t* d_seq, uint64_t total_bytes, const uint64_t* d_begins,
                                        const uint64_t* d_ends, uint64_t n_ranges, int byte, uint64_t* d_out) {
    PO_REQUIRE(ctx != nullptr, "po_count_byte_ranges_dev: ctx is NULL");
    if (n_ranges == 0) return PO_OK;
    PO_REQUIRE(d_begins && d_ends && d_out, "po_count_byte_ranges_dev: NULL buffer");
    PO_REQUIRE(d_seq != nullptr || total_bytes == 0, "po_count_byte_ranges_dev: NULL sequence buffer");
    PO_REQUIRE((reinterpret_cast<uintptr_t>(d_seq) & 15u) == 0, "po_count_byte_ranges_dev: sequence buffer must be 16-byte aligned");
    PO_REQUIRE(byte >= 0 && byte <= 255, "po_count_byte_ranges_dev: byte value %d out of range", byte);
    PO_HIP(hipSetDevice(ctx->device));
    return po_launch_count_byte_ranges(ctx, d_seq, d_begins, d_ends, n_ranges, (uint32_t)byte, d_out);
}

extern "C" int po_profile_distances_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                                        uint32_t dim, const double* d_proto, int metric, double* d_out) {
    PO_REQUIRE(ctx != nullptr, "po_profile_distances_dev: ctx is NULL");
    PO_REQUIRE(metric == PO_EUCL || metric == PO_JSD || metric == PO_KL, "po_profile_distances_dev: metric must be Eucl, JSD or KL (got %d)", metric);
    if (n == 0) return PO_OK;
    PO_REQUIRE(d_counts && d_totals && d_proto && d_out && dim > 0, "po_profile_distances_dev: bad argument");
    PO_HIP(hipSetDevice(ctx->device));
    return po_launch_profile_distances(ctx, d_counts, d_totals, n, dim, d_proto, metric, d_out);
}

extern "C" int po_profile_distances(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                                    const double* proto, int metric, double* out) {
    PO_REQUIRE(ctx != nullptr, "po_profile_distances: ctx is NULL");
    PO_REQUIRE(metric == PO_EUCL || metric == PO_JSD || metric == PO_KL, "po_profile_distances: metric must be Eucl, JSD or KL (got %d)", metric);
    if (n == 0) return PO_OK;
    PO_REQUIRE(counts && totals && proto && out && dim > 0, "po_profile_distances: bad argument");
    PO_HIP(hipSetDevice(ctx->device));
    const size_t b_cnt = po_round_up(n * (uint64_t)dim * sizeof(uint32_t), 256);
    const size_t b_tot = po_round_up(n * sizeof(uint64_t), 256);
    const size_t b_pro = po_round_up(dim * sizeof(double), 256);
    int rc = po_buf_reserve(ctx, &ctx->ws_io, b_cnt + b_tot + b_pro + n * sizeof(double));
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_io.p);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base);
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(base + b_cnt);
    double* d_pro = reinterpret_cast<double*>(base + b_cnt + b_tot);
    double* d_out = reinterpret_cast<double*>(base + b_cnt + b_tot + b_pro);
    PO_HIP(hipMemcpyAsync(d_cnt, counts, n * (uint64_t)dim * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_tot, totals, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipMemcpyAsync(d_pro, proto, dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = po_launch_profile_distances(ctx, d_cnt, d_tot, n, dim, d_pro, metric, d_out);
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    return PO_OK;
}

// ---- stage 2 -------------------------------------------------------------------------------
static const uint64_t kPadRows = 128;   // tile edge of the VALU / Gram kernels

// with_operand: also reserve the materialised Kendall pair-sign operand (up to 24 GB, FP4, folded layout).  Only
// po_pairwise_reserve asks for that - it cannot know the flags of the calls to come; a compute call reserves exactly what its
// own path uses inside po_launch_kt_pairdot_prep (PO_FLAG_NO_PAIRDOT / NO_TABLE_PATH / int8 operands / the panel kernel never
// touch this buffer, and must not fail with PO_ENOMEM because of it).  A failed reservation of the operand is not an error
// here either: the call that needs it falls back to the panel kernel.
static int reserve_pairwise(po_ctx* ctx, uint64_t n, uint32_t dim, int metric, bool with_operand) {
    const uint64_t npad = po_round_up(n ? n : 1, kPadRows);
    int rc = po_buf_reserve(ctx, &ctx->ws_rowstat, 4 * npad * sizeof(double));
    if (rc) return rc;
    if (metric != PO_KT) {
        rc = po_buf_reserve(ctx, &ctx->ws_freq, po_round_up(dim, 8) * npad * sizeof(double));
        if (rc) return rc;
    } else {
        rc = po_buf_reserve(ctx, &ctx->ws_aux, n * (uint64_t)dim * sizeof(uint32_t));
        if (rc) return rc;
        size_t kt_ws = 0;
        if (po_kt_pairdot_supported(dim)) kt_ws = po_kt_pairdot_rank_bytes(n, dim);          // ranks of the pair-dot path
        if (po_kt_panel_supported(dim) && po_kt_panel_workspace(n, dim) > kt_ws) kt_ws = po_kt_panel_workspace(n, dim);
        if (kt_ws) {
            rc = po_buf_reserve(ctx, &ctx->ws_freq, kt_ws);
            if (rc) return rc;
        }
        // the materialised pair-sign operand: sized for the reverse-complement folded layout when the word space is 4^k
        // (what `-s both` gives), else for all words; a call that needs more grows it lazily
        if (with_operand && po_kt_pairdot_supported(dim)) {
            const uint32_t selfs = po_fold_selfs(dim);
            const bool fold = selfs != 0xFFFFFFFFu;
            const size_t opb = po_kt_pairdot_operand_bytes(n, dim, fold ? selfs + (dim - selfs) / 2 : dim, fold ? selfs : 0, fold, 1);
            if (opb <= PO_PAIRDOT_MAX_OPERAND) (void)po_buf_reserve(ctx, &ctx->ws_pairdot, opb);
        }
    }
    // the pair-dot operand of an earlier Kendall / Bray-Curtis call (up to 24 GB) is dead weight for the other metrics
    if (metric != PO_KT && metric != PO_BC && ctx->ws_pairdot.cap > (1ull << 30)) {
        PO_HIP(hipStreamSynchronize(ctx->stream));
        buf_free(&ctx->ws_pairdot);
    }
    if (metric == PO_JSD) {
        rc = po_logtab_init(ctx);
        if (rc) return rc;
        rc = po_buf_reserve(ctx, &ctx->ws_aux, po_jsd_lut_workspace(n, dim));
        if (rc) return rc;
    }
    if (metric == PO_EUCL || (metric == PO_SC && po_gram_i8_sc_supported(dim))) {
        rc = po_buf_reserve(ctx, &ctx->ws_aux, po_gram_i8_workspace(n, dim));
        if (rc) return rc;
    }
    if (metric == PO_BC) {
        rc = po_buf_reserve(ctx, &ctx->ws_aux, po_bc_sad_workspace(n, dim));
        if (rc) return rc;
    }
    return PO_OK;
}

extern "C" int po_pairwise_reserve(po_ctx* ctx, uint64_t n, uint32_t dim, int metric) {
    PO_REQUIRE(ctx != nullptr, "po_pairwise_reserve: ctx is NULL");
    int rc = check_metric(metric);
    if (rc) return rc;
    PO_HIP(hipSetDevice(ctx->device));
    return reserve_pairwise(ctx, n, dim, metric, true);
}

// Exactly one of (d_counts,d_totals) / d_freq is given.  Prepares the working layout once, then
// launches every block.
static int pairwise_core(po_ctx* ctx, const char* who, const uint32_t* d_counts, const uint64_t* d_totals,
                         const double* d_freq, uint64_t n, uint32_t dim, int metric, int out_dtype,
                         const po_block* blocks, uint32_t n_blocks, uint32_t flags, po_stats* stats) {
    PO_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    int rc = check_metric(metric);
    if (rc) return rc;
    PO_REQUIRE(out_dtype == PO_F64 || out_dtype == PO_F32, "%s: out_dtype must be PO_F64 or PO_F32", who);
    PO_REQUIRE(dim >= 1, "%s: dim must be positive", who);
    PO_REQUIRE(blocks != nullptr || n_blocks == 0, "%s: NULL block list", who);
    if (stats) memset(stats, 0, sizeof(*stats));
    uint64_t entries = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const po_block& k = blocks[b];
        PO_REQUIRE(k.row_begin <= k.row_end && k.row_end <= n && k.col_begin <= k.col_end && k.col_end <= n,
                   "%s: block %u [%llu,%llu) x [%llu,%llu) outside 0..%llu", who, b, (unsigned long long)k.row_begin,
                   (unsigned long long)k.row_end, (unsigned long long)k.col_begin, (unsigned long long)k.col_end,
                   (unsigned long long)n);
        if (k.row_begin == k.row_end || k.col_begin == k.col_end) continue;
        PO_REQUIRE(k.out != nullptr, "%s: block %u has no output buffer", who, b);
        PO_REQUIRE(k.ld_out >= k.col_end - k.col_begin, "%s: block %u: ld_out (%llu) < columns (%llu)", who, b,
                   (unsigned long long)k.ld_out, (unsigned long long)(k.col_end - k.col_begin));
        if (k.triangular)
            PO_REQUIRE(k.row_begin == k.col_begin && k.row_end == k.col_end, "%s: block %u is triangular but rows != columns", who, b);
        else if (k.mirror)
            PO_REQUIRE(k.ld_mirror >= k.row_end - k.row_begin, "%s: block %u: ld_mirror (%llu) < rows (%llu)", who, b,
                       (unsigned long long)k.ld_mirror, (unsigned long long)(k.row_end - k.row_begin));
        entries += (k.row_end - k.row_begin) * (k.col_end - k.col_begin) * ((k.mirror && !k.triangular) ? 2 : 1);
    }
    if (n == 0 || entries == 0) return PO_OK;
    PO_REQUIRE(d_freq != nullptr || (d_counts != nullptr && d_totals != nullptr), "%s: NULL buffer", who);
    PO_HIP(hipSetDevice(ctx->device));

    rc = reserve_pairwise(ctx, n, dim, metric, false);
    if (rc) return rc;
    const uint64_t npad = po_round_up(n, kPadRows);
    double* ft = static_cast<double*>(ctx->ws_freq.p);
    double* rowstat = static_cast<double*>(ctx->ws_rowstat.p);

    if (stats) PO_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    // ---- frequencies that are count2freq output: back to the exact integer profiles (po_recover.hip) ----
    if (d_freq && !(flags & PO_FLAG_NO_TABLE_PATH)) {
        bool recovered = false;
        const uint32_t* rc_counts = nullptr;
        const uint64_t* rc_totals = nullptr;
        rc = po_recover_counts(ctx, d_freq, n, dim, &recovered, &rc_counts, &rc_totals);
        if (rc) return rc;
        if (recovered) {
            d_counts = rc_counts;
            d_totals = rc_totals;
            d_freq = nullptr;
        }
    }
    // ---- strand-symmetric profiles: one word per reverse-complement orbit (JSD, BC; po_fold.hip) ----
    uint32_t dbl_at = PO_NO_DOUBLING;
    bool folded = false;
    uint32_t fold_flags = 0xFFFFFFFFu;     // all set = unknown: launch every candidate kernel
    if ((metric == PO_JSD || metric == PO_BC) && !(flags & PO_FLAG_NO_RC_FOLD)) {
        uint32_t dim_f = 0, at = PO_NO_DOUBLING;
        rc = po_rc_fold(ctx, d_counts, d_freq, d_totals, n, dim, metric == PO_BC ? 32u : 8u, &folded, &dim_f, &at, &fold_flags);
        if (rc) return rc;
        if (folded) {
            if (d_counts) d_counts = static_cast<const uint32_t*>(ctx->ws_fold.p);
            else d_freq = static_cast<const double*>(ctx->ws_fold.p);
            dim = dim_f;
            dbl_at = at;
        }
    }
    // ---- prep: working layout + per-row terms (once) ----
    uint32_t* lessrank = nullptr;
    const unsigned long long* cls = nullptr;
    const uint32_t* i8flag = nullptr;
    if (metric == PO_EUCL && d_counts && !(flags & PO_FLAG_NO_TABLE_PATH)) {
        // profiles that fit int8 go through the exact integer MFMA kernel; both tile kernels are launched
        // and the device-side flag decides which one does the work (and whether the float64 operand
        // matrix is built at all)
        rc = po_launch_gram_i8_prep(ctx, d_counts, d_totals, false, n, dim, npad, ctx->ws_aux.p, &i8flag);
        if (rc) return rc;
    }
    const uint32_t i8_upto = po_gram_i8_value_limit(dim);
    // Spearman: the doubled centred ranks are small integers -> the same exact int8 kernel (two digit planes)
    const bool sc_i8 = metric == PO_SC && po_gram_i8_sc_supported(dim) && !(flags & PO_FLAG_NO_TABLE_PATH);
    // the fold pass may have established that the equal-total kernels own every tile: the float64 operand matrix
    // is then not needed at all and the per-record terms come straight from the counts
    const bool all_table = d_counts && !(flags & PO_FLAG_NO_TABLE_PATH) &&
                           ((metric == PO_JSD && !(fold_flags & PO_FOLD_NOT_ALL_TABLE)) ||
                            (metric == PO_BC && !(fold_flags & PO_FOLD_NOT_ALL_SAD)));
    if ((metric == PO_EUCL || metric == PO_JSD || metric == PO_BC) && !all_table) {
        rc = d_freq ? po_launch_prep_freq(ctx, d_freq, n, dim, npad, ft)
                    : po_launch_prep(ctx, d_counts, d_totals, n, dim, npad, ft, i8flag, i8_upto);
        if (rc) return rc;
    }
    if (all_table) {
        rc = po_launch_rowstat_counts(ctx, d_counts, d_totals, n, dim, npad, rowstat, metric == PO_JSD, dbl_at);
    } else if (metric == PO_JSD || metric == PO_BC) {
        rc = po_launch_rowstat(ctx, ft, n, dim, npad, rowstat, metric == PO_JSD ? ctx->ws_logtab.p : nullptr, dbl_at);
    } else if (sc_i8) {
        int32_t* r2 = static_cast<int32_t*>(ctx->ws_freq.p);           // the float64 operand buffer is free on this path
        rc = po_launch_ranks(ctx, d_freq ? nullptr : d_counts, d_freq, n, dim, npad, nullptr, nullptr, r2, rowstat);
        if (rc) return rc;
        rc = po_launch_gram_i8_prep(ctx, reinterpret_cast<const uint32_t*>(r2), nullptr, true, n, dim, npad, ctx->ws_aux.p, nullptr);
    } else if (metric == PO_SC) {
        rc = po_launch_ranks(ctx, d_freq ? nullptr : d_counts, d_freq, n, dim, npad, ft, nullptr, nullptr, rowstat);
    } else if (metric == PO_KT) {
        lessrank = static_cast<uint32_t*>(ctx->ws_aux.p);
        rc = po_launch_ranks(ctx, d_freq ? nullptr : d_counts, d_freq, n, dim, npad, nullptr, lessrank, nullptr, rowstat);
    }
    if (rc) return rc;
    po_pairdot_plan kt_plan, bc_plan;
    bool bc_thermo = false;
    double bc_inv_n = 0.0;
    memset(&bc_plan, 0, sizeof(bc_plan));
    po_kt_panel_plan kt_pplan;
    memset(&kt_plan, 0, sizeof(kt_plan));
    memset(&kt_pplan, 0, sizeof(kt_pplan));
    bool kt_mfma = metric == PO_KT && po_kt_pairdot_supported(dim) && !(flags & (PO_FLAG_NO_TABLE_PATH | PO_FLAG_NO_PAIRDOT));
    const bool kt_panel_ok = metric == PO_KT && po_kt_panel_supported(dim) && !(flags & PO_FLAG_NO_TABLE_PATH);
    bool kt_panel = false;
    if (kt_mfma || kt_panel_ok) {
        // strand-symmetric records: Kendall's S over one word per reverse-complement orbit, weighted (po_fold.hip)
        const uint32_t selfs = po_fold_selfs(dim);
        const uint32_t* fold_src = nullptr;
        uint32_t fold_len = 0;
        const bool can_fold = selfs != 0xFFFFFFFFu && (kt_mfma || po_kt_panel_fold_supported(dim, selfs));
        if (!(flags & PO_FLAG_NO_RC_FOLD) && can_fold) {
            uint32_t at = 0;
            rc = po_rc_fold(ctx, d_counts, d_freq, nullptr, n, dim, PO_FOLD_SELFS_FIRST, &folded, &fold_len, &at, nullptr);
            if (rc) return rc;
            if (folded) fold_src = static_cast<const uint32_t*>(ctx->ws_fold_src.p);
        }
        const uint32_t n_pairs = fold_src ? (dim - selfs) / 2 : 0;
        const int want_fp4 = (flags & PO_FLAG_PAIRDOT_I8) ? 0 : 1;
        // the materialised operand must stay within bounds (it grows with D^2: 1 MB per record at folded k = 6)
        if (kt_mfma && po_kt_pairdot_operand_bytes(n, dim, fold_src ? selfs + n_pairs : dim, selfs, fold_src != nullptr, want_fp4) >
                           PO_PAIRDOT_MAX_OPERAND)
            kt_mfma = false;
        if (kt_mfma) {
            rc = po_launch_kt_pairdot_prep(ctx, lessrank, n, dim, npad, fold_src, selfs, n_pairs, want_fp4, &kt_plan);
            if (rc == PO_ENOMEM) { rc = PO_OK; kt_mfma = false; }    // no room for the operand: the panel / VALU kernel does without
        }
        kt_panel = !kt_mfma && kt_panel_ok;
        if (kt_panel && fold_src && !po_kt_panel_fold_supported(dim, selfs)) { fold_src = nullptr; folded = false; }
        if (kt_panel) rc = po_launch_kt_panel_prep(ctx, lessrank, n, dim, npad, ctx->ws_freq.p, fold_src, fold_len, selfs, n_pairs, &kt_pplan);
        if (rc) return rc;
    }
    if (metric == PO_EUCL || (metric == PO_SC && !sc_i8)) {
        rc = po_launch_gram_norms(ctx, ft, dim, npad, rowstat, i8flag, i8_upto);
        if (rc) return rc;
    }
    if (metric == PO_JSD && d_counts && !(flags & PO_FLAG_NO_TABLE_PATH)) {
        // record blocks with one common word total go through the integer-sum table kernel, the rest
        // through the general float64 kernel; each launch skips the other's tiles (decided on device)
        rc = po_launch_jsd_lut_prep(ctx, d_counts, d_totals, n, dim, npad, rowstat + npad, ctx->ws_aux.p, &cls);
        if (rc) return rc;
    }
    if (metric == PO_BC && d_counts && !(flags & PO_FLAG_NO_TABLE_PATH)) {
        // same split for Bray-Curtis: equal-total record blocks with byte-sized counts take the packed SAD kernel
        rc = po_launch_bc_sad_prep(ctx, d_counts, d_totals, n, dim, npad, ctx->ws_aux.p, &cls);
        if (rc) return rc;
        // one common word total and few count levels per word: sum of min as a matrix-core Gram over thermometer
        // planes (po_pairdot.hip); decided from a small header read back from the device
        const uint32_t* p8t = nullptr;
        uint32_t gp = 0;
        po_bc_sad_view(ctx->ws_aux.p, npad, dim, &p8t, &gp);
        if (!(flags & PO_FLAG_NO_PAIRDOT))
            rc = po_launch_bc_thermo_prep(ctx, p8t, gp, cls, n, dim, npad, dbl_at, (flags & PO_FLAG_PAIRDOT_I8) ? 0 : 1,
                                      &bc_thermo, &bc_plan, &bc_inv_n);
        if (rc) return rc;
    }
    if (stats) PO_HIP(hipEventRecord(ctx->ev[1], ctx->stream));

    // ---- tiles, block by block ----
    uint64_t tiles = 0;
    uint32_t kid = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const po_block& k = blocks[b];
        if (k.row_begin == k.row_end || k.col_begin == k.col_end) continue;
        po_tile_args a;
        a.ft = ft;
        a.rowstat = rowstat;
        a.n = n;
        a.npad = npad;
        a.dim = dim;
        a.row_begin = k.row_begin; a.row_end = k.row_end;
        a.col_begin = k.col_begin; a.col_end = k.col_end;
        a.out = k.out;
        a.ld_out = k.ld_out;
        a.triangular = k.triangular ? 1 : 0;
        a.mirror = k.triangular ? k.out : k.mirror;
        a.ld_mirror = k.triangular ? k.ld_out : k.ld_mirror;
        a.out_f32 = (out_dtype == PO_F32);
        a.dbl_at = dbl_at;
        switch (metric) {
            case PO_JSD:
                if (cls) {
                    rc = po_launch_jsd_lut_tiles(ctx, a, n, ctx->ws_aux.p, &tiles);
                    if (rc) return rc;
                }
                // the fold pass may already have established that every record block takes the table kernel
                if (!(cls && !(fold_flags & PO_FOLD_NOT_ALL_TABLE))) rc = po_launch_valu_tiles(ctx, PO_JSD, a, cls, cls ? nullptr : &tiles);
                kid = cls ? PO_KERNEL_LUT_JSD : PO_KERNEL_VALU_JSD;
                break;
            case PO_BC:
                if (bc_thermo) {
                    rc = po_launch_bc_thermo_tiles(ctx, a, bc_plan, bc_inv_n, &tiles);
                    kid = PO_KERNEL_MFMA_BC;
                    break;
                }
                if (cls) {
                    rc = po_launch_bc_sad_tiles(ctx, a, ctx->ws_aux.p, &tiles);
                    if (rc) return rc;
                }
                if (!(cls && !(fold_flags & PO_FOLD_NOT_ALL_SAD))) rc = po_launch_valu_tiles(ctx, PO_BC, a, cls, cls ? nullptr : &tiles);
                kid = cls ? PO_KERNEL_SAD_BC : PO_KERNEL_VALU_BC;
                break;
            case PO_EUCL:
                if (i8flag) {
                    rc = po_launch_gram_i8_tiles(ctx, PO_EUCL, a, ctx->ws_aux.p, &tiles);
                    if (rc) return rc;
                }
                rc = po_launch_gram_f64(ctx, PO_EUCL, a, i8flag, i8_upto, i8flag ? nullptr : &tiles);
                kid = i8flag ? PO_KERNEL_MFMA_I8_GRAM : PO_KERNEL_MFMA_F64_GRAM;
                break;
            case PO_SC:
                if (sc_i8) { rc = po_launch_gram_i8_tiles(ctx, PO_SC, a, ctx->ws_aux.p, &tiles); kid = PO_KERNEL_MFMA_I8_GRAM; }
                else { rc = po_launch_gram_f64(ctx, PO_SC, a, nullptr, 0, &tiles); kid = PO_KERNEL_MFMA_F64_GRAM; }
                break;
            case PO_KT:
                if (kt_mfma) { rc = po_launch_kt_pairdot_tiles(ctx, a, kt_plan, &tiles); kid = PO_KERNEL_MFMA_I8_KT; }
                else if (kt_panel) { rc = po_launch_kt_panel_tiles(ctx, a, ctx->ws_freq.p, kt_pplan, &tiles); kid = PO_KERNEL_MFMA_I8_KT; }
                else { rc = po_launch_kt(ctx, lessrank, n, dim, a, &tiles); kid = PO_KERNEL_VALU_KT; }
                break;
        }
        if (rc) return rc;
    }
    if (stats) {
        PO_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
        PO_HIP(hipEventSynchronize(ctx->ev[2]));
        float ms = 0.f;
        PO_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        stats->prep_ms = ms;
        PO_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
        stats->kernel_ms = ms;
        PO_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[2]));
        stats->total_ms = ms;
        stats->pairs = entries / 2;
        stats->tiles = tiles;
        stats->kernel_id = kid;
        stats->rc_folded = folded ? 1u : 0u;
    }
    return PO_OK;
}

// rows [row_begin,row_end) x all columns as one block; the full matrix is one triangular block
static po_block rows_block(uint64_t n, uint64_t row_begin, uint64_t row_end, void* out, uint64_t ld_out, uint32_t flags) {
    po_block k;
    memset(&k, 0, sizeof(k));
    k.row_begin = row_begin; k.row_end = row_end;
    k.col_begin = 0; k.col_end = n;
    k.out = out; k.ld_out = ld_out;
    k.triangular = (row_begin == 0 && row_end == n && !(flags & PO_FLAG_NO_SYMMETRY)) ? 1u : 0u;
    return k;
}

static int check_rows(const char* who, uint64_t n, uint64_t row_begin, uint64_t row_end, const void* out, uint64_t ld_out) {
    PO_REQUIRE(row_begin <= row_end && row_end <= n, "%s: row range [%llu,%llu) outside 0..%llu", who,
               (unsigned long long)row_begin, (unsigned long long)row_end, (unsigned long long)n);
    if (n == 0 || row_begin == row_end) return PO_OK;
    PO_REQUIRE(out != nullptr, "%s: NULL buffer", who);
    PO_REQUIRE(ld_out >= n, "%s: ld_out (%llu) < n (%llu)", who, (unsigned long long)ld_out, (unsigned long long)n);
    return PO_OK;
}

extern "C" int po_pairwise_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                               uint32_t dim, int metric, uint64_t row_begin, uint64_t row_end, int out_dtype,
                               void* d_out, uint64_t ld_out, uint32_t flags, po_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    int rc = check_rows("po_pairwise_dev", n, row_begin, row_end, d_out, ld_out);
    if (rc) return rc;
    if (n && row_begin != row_end && (!d_counts || !d_totals)) { po_set_error("po_pairwise_dev: NULL buffer"); return PO_EINVAL; }
    const po_block k = rows_block(n, row_begin, row_end, d_out, ld_out, flags);
    return pairwise_core(ctx, "po_pairwise_dev", d_counts, d_totals, nullptr, n, dim, metric, out_dtype, &k, 1, flags, stats);
}

extern "C" int po_pairwise_freq_dev(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, int metric,
                                    uint64_t row_begin, uint64_t row_end, int out_dtype, void* d_out,
                                    uint64_t ld_out, uint32_t flags, po_stats* stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    int rc = check_rows("po_pairwise_freq_dev", n, row_begin, row_end, d_out, ld_out);
    if (rc) return rc;
    if (n && row_begin != row_end && !d_freq) { po_set_error("po_pairwise_freq_dev: NULL buffer"); return PO_EINVAL; }
    const po_block k = rows_block(n, row_begin, row_end, d_out, ld_out, flags);
    return pairwise_core(ctx, "po_pairwise_freq_dev", nullptr, nullptr, d_freq, n, dim, metric, out_dtype, &k, 1, flags, stats);
}

extern "C" int po_pairwise_blocks_dev(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                                      uint32_t dim, int metric, int out_dtype, const po_block* blocks,
                                      uint32_t n_blocks, uint32_t flags, po_stats* stats) {
    return pairwise_core(ctx, "po_pairwise_blocks_dev", d_counts, d_totals, nullptr, n, dim, metric, out_dtype, blocks,
                         n_blocks, flags, stats);
}

// Device rows -> pageable host rows through two pinned staging buffers: the DMA of chunk c+1 runs while host
// threads copy chunk c into the caller's memory.  Measured on the gpurun box for 6 GB (tools/ubench/d2h_paths.hip):
// plain hipMemcpy into fresh pageable memory 18 GB/s; hipHostRegister of the destination + copy 15-17 GB/s all
// in (registering costs 0.3 s); hipHostMalloc of a pinned result 7 GB/s all in (0.8 s to allocate, 0.5 s to free);
// this ring into 4 KiB pages 15 GB/s, into transparent huge pages 43 GB/s (the link itself does 57 GB/s).
static int copy_rows_to_host(po_ctx* ctx, const uint8_t* d_src, size_t src_pitch, size_t row_bytes, uint64_t rows,
                             uint8_t* dst, size_t dst_pitch) {
    const size_t kStage = 32u << 20;
    if (rows == 0 || row_bytes == 0) return PO_OK;
    if (rows * row_bytes < (64u << 20) || row_bytes > kStage) {      // small result (or absurdly long rows): one plain copy
        PO_HIP(hipMemcpy2DAsync(dst, dst_pitch, d_src, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
        PO_HIP(hipStreamSynchronize(ctx->stream));
        return PO_OK;
    }
    // A freshly allocated destination (numpy.zeros / numpy.empty: untouched anonymous pages) is first touched by the
    // copy threads below; with 4 KiB pages that is 1.8 million page faults for a 7 GB matrix and caps the copy at
    // ~15 GB/s.  Asking for transparent huge pages first (harmless if the range is already populated or the kernel
    // declines) measured 43 GB/s on the same box (tools/ubench/d2h_paths.hip: ring 2 x 32 MB, 8 threads, THP).
    {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + 4095u) & ~(uintptr_t)4095u;
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + (rows - 1) * dst_pitch + row_bytes) & ~(uintptr_t)4095u;
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
    if (!ctx->h_stage[0]) {
        PO_HIP(hipHostMalloc(&ctx->h_stage[0], kStage, hipHostMallocDefault));
        PO_HIP(hipHostMalloc(&ctx->h_stage[1], kStage, hipHostMallocDefault));
    }
    // the ring itself is host code (po_ring_copy_rows, po_io.cpp: it also runs under the thread sanitizer, `make san`);
    // here only the two HIP calls it is built around
    struct d2h_src { po_ctx* ctx; const uint8_t* d_src; size_t src_pitch, row_bytes; hipError_t err; } u{ctx, d_src, src_pitch, row_bytes, hipSuccess};
    po_ring_source ring;
    ring.user = &u;
    ring.issue = [](void* p, uint64_t, void* stage, uint64_t r0, uint64_t nr) -> int {
        d2h_src* s = static_cast<d2h_src*>(p);
        s->err = hipMemcpy2DAsync(stage, s->row_bytes, s->d_src + r0 * s->src_pitch, s->src_pitch, s->row_bytes, nr,
                                  hipMemcpyDeviceToHost, s->ctx->stream);
        return s->err == hipSuccess ? PO_OK : PO_EHIP;
    };
    ring.wait = [](void* p) -> int {
        d2h_src* s = static_cast<d2h_src*>(p);
        s->err = hipStreamSynchronize(s->ctx->stream);
        return s->err == hipSuccess ? PO_OK : PO_EHIP;
    };
    const int rc = po_ring_copy_rows(ring, ctx->h_stage, kStage, row_bytes, rows, dst, dst_pitch, po_host_threads(14));
    if (rc == PO_EHIP) po_set_error("device to host copy: %s", hipGetErrorString(u.err));
    return rc;
}

// host-pointer forms: stage in ws_io, run the device form, copy the rows back
static int pairwise_host(po_ctx* ctx, const char* who, const uint32_t* counts, const uint64_t* totals,
                         const double* freq, uint64_t n, uint32_t dim, int metric, uint64_t row_begin,
                         uint64_t row_end, int out_dtype, void* out, uint64_t ld_out, uint32_t flags,
                         po_stats* stats) {
    PO_REQUIRE(ctx != nullptr, "%s: ctx is NULL", who);
    int rc = check_metric(metric);
    if (rc) return rc;
    PO_REQUIRE(out_dtype == PO_F64 || out_dtype == PO_F32, "%s: out_dtype must be PO_F64 or PO_F32", who);
    PO_REQUIRE(row_begin <= row_end && row_end <= n, "%s: row range [%llu,%llu) outside 0..%llu", who,
               (unsigned long long)row_begin, (unsigned long long)row_end, (unsigned long long)n);
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n == 0 || row_begin == row_end) return PO_OK;
    PO_REQUIRE((freq != nullptr || (counts != nullptr && totals != nullptr)) && out != nullptr, "%s: NULL buffer", who);
    PO_REQUIRE(ld_out >= n, "%s: ld_out (%llu) < n (%llu)", who, (unsigned long long)ld_out, (unsigned long long)n);
    PO_REQUIRE(dim >= 1, "%s: dim must be positive", who);
    PO_HIP(hipSetDevice(ctx->device));

    const size_t esz = (out_dtype == PO_F32) ? 4 : 8;
    const uint64_t rows = row_end - row_begin;
    const size_t b_in = po_round_up(n * (uint64_t)dim * (freq ? sizeof(double) : sizeof(uint32_t)), 256);
    const size_t b_tot = po_round_up(n * sizeof(uint64_t), 256);
    const size_t b_out = rows * n * esz;
    rc = po_buf_reserve(ctx, &ctx->ws_io, b_in + b_tot + b_out);
    if (rc) return rc;
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_io.p);
    void* d_in = base;
    uint64_t* d_tot = reinterpret_cast<uint64_t*>(base + b_in);
    void* d_out = base + b_in + b_tot;
    if (freq) {
        PO_HIP(hipMemcpyAsync(d_in, freq, n * (uint64_t)dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const po_block k = rows_block(n, row_begin, row_end, d_out, n, flags);
        rc = pairwise_core(ctx, who, nullptr, nullptr, static_cast<const double*>(d_in), n, dim, metric, out_dtype, &k, 1,
                           flags, stats);
    } else {
        PO_HIP(hipMemcpyAsync(d_in, counts, n * (uint64_t)dim * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PO_HIP(hipMemcpyAsync(d_tot, totals, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        const po_block k = rows_block(n, row_begin, row_end, d_out, n, flags);
        rc = pairwise_core(ctx, who, static_cast<const uint32_t*>(d_in), d_tot, nullptr, n, dim, metric, out_dtype, &k, 1,
                           flags, stats);
    }
    if (rc) return rc;
    return copy_rows_to_host(ctx, static_cast<const uint8_t*>(d_out), n * esz, n * esz, rows, static_cast<uint8_t*>(out), ld_out * esz);
}

extern "C" int po_pairwise(po_ctx* ctx, const uint32_t* counts, const uint64_t* totals, uint64_t n, uint32_t dim,
                           int metric, uint64_t row_begin, uint64_t row_end, int out_dtype, void* out,
                           uint64_t ld_out, uint32_t flags, po_stats* stats) {
    if (n && row_begin != row_end && (!counts || !totals)) { po_set_error("po_pairwise: NULL buffer"); return PO_EINVAL; }
    return pairwise_host(ctx, "po_pairwise", counts, totals, nullptr, n, dim, metric, row_begin, row_end, out_dtype,
                         out, ld_out, flags, stats);
}

extern "C" int po_pairwise_freq(po_ctx* ctx, const double* freq, uint64_t n, uint32_t dim, int metric,
                                uint64_t row_begin, uint64_t row_end, int out_dtype, void* out, uint64_t ld_out,
                                uint32_t flags, po_stats* stats) {
    if (n && row_begin != row_end && !freq) { po_set_error("po_pairwise_freq: NULL buffer"); return PO_EINVAL; }
    return pairwise_host(ctx, "po_pairwise_freq", nullptr, nullptr, freq, n, dim, metric, row_begin, row_end,
                         out_dtype, out, ld_out, flags, stats);
}
