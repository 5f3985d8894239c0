// FASTA ingest on the device: the raw file bytes are in HBM, the kernels below produce what po_fasta_scan /
// po_fasta_extract (po_io.cpp) produce on the host - the concatenated sequence bytes, the record offsets and the
// title spans - without the file ever being walked by a CPU.
//
// Reference: records come from Bio.SeqIO.parse(genome, "fasta") (/root/reference/phylopackage/bin/phyloligo.py:869);
// Biopython is third-party and absent from the reference tree, the semantics are those restated in po_io.cpp
// (SimpleFastaParser): a line whose first byte is '>' opens a record, the title is the rest of that line, the
// sequence is the following lines right-stripped and joined with every ' ' and '\r' removed; text before the first
// record must be blank.
//
// Byte-parallel formulation.  For byte i let LS(i) be the start of its line (the position after the last '\n' before
// i, or 0).  Then
//     header(i)  =  data[LS(i)] == '>'
//     keep(i)    =  i >= F  and  not header(i)  and  data[i] not in {' ', '\r', '\n'}        F = first header line start
// and the output position of a kept byte is the number of kept bytes before it; record r starts at the r-th header
// line and its offset is the number of kept bytes before that line.  LS is a running maximum, the positions are
// running sums: block-local scans in LDS (4 KiB of file per 256-lane workgroup, 16 bytes per lane) + one small scan
// over the per-block partials.  Three passes over the file for the sizes, one more to write.
// One construct is left to the host parser: a tab / vertical tab / form feed on a sequence line is kept by rstrip()
// unless only white space follows it on the line, which needs a backward scan; such files (none seen in practice) make
// po_fasta_scan_dev return PO_EUNSUPPORTED.  Title spans end at the line end; trailing white space is stripped by
// whoever decodes a title (the host layer does, lazily).
#include "po_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPerLane = 16;
constexpr int kBlockBytes = kThreads * kPerLane;        // 4 KiB of file per workgroup
constexpr long long kNone = -1;

struct fasta_totals {            // device + pinned host copy
    unsigned long long first_header;   // F (len if the file has no record)
    unsigned long long n_records, seq_bytes;
    unsigned long long junk_before;    // non-blank bytes before F
    unsigned long long odd_space;      // '\t' '\v' '\f' on sequence lines
};

__device__ __forceinline__ bool py_space(uint32_t c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c;
}

// the lane's 16 bytes (zero beyond the file) and the byte before them ('\n' before the file: position 0 is a line start)
__device__ __forceinline__ void load_chunk(const uint8_t* __restrict__ data, uint64_t len, uint64_t p0, uint8_t (&c)[kPerLane], uint32_t& prev) {
    if (p0 + kPerLane <= len) {
        const uint4 v = *reinterpret_cast<const uint4*>(data + p0);      // d_data is 16-byte aligned, p0 a multiple of 16
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) c[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) c[i] = (p0 + i < len) ? data[p0 + i] : (uint8_t)0;
    }
    prev = (p0 == 0) ? (uint32_t)'\n' : (p0 - 1 < len ? (uint32_t)data[p0 - 1] : 0u);
}

// pass 1: last line start inside every 4 KiB block (kNone if none) and the first header line of the file
__global__ __launch_bounds__(kThreads) void fasta_marks_kernel(const uint8_t* __restrict__ data, uint64_t len,
                                                               long long* __restrict__ blk_last_start,
                                                               unsigned long long* __restrict__ first_header) {
    __shared__ long long red[kThreads / 64];
    __shared__ unsigned long long redf[kThreads / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * kBlockBytes + (uint64_t)threadIdx.x * kPerLane;
    uint8_t c[kPerLane];
    uint32_t prev;
    load_chunk(data, len, p0, c, prev);
    long long last = kNone;
    unsigned long long fh = ~0ull;
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
        const uint64_t p = p0 + i;
        if (p < len && prev == '\n') {
            last = (long long)p;
            if (c[i] == '>' && fh == ~0ull) fh = p;
        }
        prev = c[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        last = max(last, (long long)__shfl_down(last, o, 64));
        const unsigned long long f2 = __shfl_down(fh, o, 64);
        fh = f2 < fh ? f2 : fh;
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = last; redf[threadIdx.x >> 6] = fh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long m = red[0];
        unsigned long long f = redf[0];
        for (int w = 1; w < kThreads / 64; ++w) { m = max(m, red[w]); f = redf[w] < f ? redf[w] : f; }
        blk_last_start[blockIdx.x] = m;
        if (f != ~0ull) atomicMin(first_header, f);
    }
}

// exclusive running maximum over the per-block values (one workgroup; 1024 blocks = 4 MiB of file per round)
__global__ __launch_bounds__(1024) void scan_max_kernel(const long long* __restrict__ in, uint64_t nb, long long* __restrict__ out) {
    __shared__ long long part[1024];
    const uint32_t t = threadIdx.x;
    long long carry = kNone;
    for (uint64_t b0 = 0; b0 < nb; b0 += 1024) {
        const long long v = (b0 + t < nb) ? in[b0 + t] : kNone;
        part[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const long long u = (t >= d) ? part[t - d] : kNone;
            __syncthreads();
            part[t] = max(part[t], u);
            __syncthreads();
        }
        if (b0 + t < nb) out[b0 + t] = max(carry, t ? part[t - 1] : kNone);      // exclusive
        const long long round_max = part[1023];
        __syncthreads();
        carry = max(carry, round_max);
    }
}

// exclusive running sums of two per-block counters; totals[0], totals[1] = their grand totals
__global__ __launch_bounds__(1024) void scan_sum2_kernel(unsigned long long* __restrict__ a, unsigned long long* __restrict__ b,
                                                         uint64_t nb, unsigned long long* __restrict__ tot_a,
                                                         unsigned long long* __restrict__ tot_b) {
    __shared__ unsigned long long pa[1024], pb[1024];
    const uint32_t t = threadIdx.x;
    unsigned long long ca = 0, cb = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += 1024) {
        const unsigned long long va = (b0 + t < nb) ? a[b0 + t] : 0ull, vb = (b0 + t < nb) ? b[b0 + t] : 0ull;
        pa[t] = va; pb[t] = vb;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const unsigned long long ua = (t >= d) ? pa[t - d] : 0ull, ub = (t >= d) ? pb[t - d] : 0ull;
            __syncthreads();
            pa[t] += ua; pb[t] += ub;
            __syncthreads();
        }
        if (b0 + t < nb) { a[b0 + t] = ca + pa[t] - va; b[b0 + t] = cb + pb[t] - vb; }
        const unsigned long long ra = pa[1023], rb = pb[1023];
        __syncthreads();
        ca += ra; cb += rb;
    }
    if (t == 0) { *tot_a = ca; *tot_b = cb; }
}

// What the walk of one lane's 16 bytes needs to know at its first byte: the start of the line it is in.  The lanes'
// own last line starts are combined by an exclusive running maximum in LDS, on top of the block's carry-in.
__device__ __forceinline__ long long line_start_before(long long own_last, long long carry_in, long long* scratch) {
    const uint32_t t = threadIdx.x;
    scratch[t] = own_last;
    __syncthreads();
    for (uint32_t d = 1; d < kThreads; d <<= 1) {
        const long long u = (t >= d) ? scratch[t - d] : kNone;
        __syncthreads();
        scratch[t] = max(scratch[t], u);
        __syncthreads();
    }
    const long long before = max(carry_in, t ? scratch[t - 1] : kNone);
    __syncthreads();
    return before;
}

// WRITE = false (pass 2): per-block counts of kept bytes and of header lines, file-level checks.
// WRITE = true  (pass 4): the kept bytes to their places, offsets and title spans of the records.
template <bool WRITE>
__global__ __launch_bounds__(kThreads) void fasta_walk_kernel(const uint8_t* __restrict__ data, uint64_t len,
                                                              const long long* __restrict__ carry_ls,
                                                              const fasta_totals* __restrict__ tot,
                                                              unsigned long long* __restrict__ blk_keep,
                                                              unsigned long long* __restrict__ blk_hdr,
                                                              fasta_totals* __restrict__ tot_out,
                                                              uint8_t* __restrict__ seq_out, unsigned long long* __restrict__ offsets,
                                                              unsigned long long* __restrict__ title_begin,
                                                              unsigned long long* __restrict__ title_end) {
    __shared__ long long scratch[kThreads];
    __shared__ unsigned long long sk[kThreads], sh[kThreads];
    const uint32_t t = threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * kBlockBytes + (uint64_t)t * kPerLane;
    const unsigned long long F = tot->first_header;
    uint8_t c[kPerLane];
    uint32_t prev0;
    load_chunk(data, len, p0, c, prev0);
    // the lane's own last line start
    long long own_last = kNone;
    {
        uint32_t prev = prev0;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            if (p0 + i < len && prev == '\n') own_last = (long long)(p0 + i);
            prev = c[i];
        }
    }
    const long long ls0 = line_start_before(own_last, carry_ls[blockIdx.x], scratch);
    bool header = ls0 >= 0 && data[ls0] == '>';                       // the line the lane's first byte belongs to
    // ---- walk: counts (and, when writing, the local positions) ----
    uint32_t keepmask = 0, hdrmask = 0;
    unsigned long long junk = 0, odd = 0;
    {
        uint32_t prev = prev0;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const uint64_t p = p0 + i;
            if (p < len) {
                if (prev == '\n') {
                    header = c[i] == '>';
                    if (header) hdrmask |= 1u << i;
                }
                const uint32_t ch = c[i];
                if (p < F) {
                    junk += py_space(ch) ? 0u : 1u;
                } else if (!header && ch != ' ' && ch != '\r' && ch != '\n') {
                    keepmask |= 1u << i;
                    odd += (ch == '\t' || ch == 0x0b || ch == 0x0c) ? 1u : 0u;
                }
            }
            prev = c[i];
        }
    }
    const uint32_t nk = __popc(keepmask), nh = __popc(hdrmask);
    // exclusive sums over the lanes of the block
    sk[t] = nk; sh[t] = nh;
    __syncthreads();
    for (uint32_t d = 1; d < kThreads; d <<= 1) {
        const unsigned long long uk = (t >= d) ? sk[t - d] : 0ull, uh = (t >= d) ? sh[t - d] : 0ull;
        __syncthreads();
        sk[t] += uk; sh[t] += uh;
        __syncthreads();
    }
    if (!WRITE) {
        if (t == kThreads - 1) { blk_keep[blockIdx.x] = sk[t]; blk_hdr[blockIdx.x] = sh[t]; }
        if (junk) atomicAdd(&tot_out->junk_before, junk);
        if (odd) atomicAdd(&tot_out->odd_space, odd);
        return;
    }
    unsigned long long kpos = blk_keep[blockIdx.x] + sk[t] - nk;       // kept bytes before the lane's first byte
    unsigned long long hidx = blk_hdr[blockIdx.x] + sh[t] - nh;        // header lines before it
    // is the lane's first line a header whose '\n' this lane may hold?  `header` was left at its end-of-walk value above
    bool hdr_line = ls0 >= 0 && data[ls0] == '>';
    uint32_t prev = prev0;
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
        const uint64_t p = p0 + i;
        if (p < len) {
            if (prev == '\n') {
                hdr_line = c[i] == '>';
                if (hdr_line) {
                    offsets[hidx] = kpos;
                    title_begin[hidx] = p + 1;
                    ++hidx;
                }
            }
            if (c[i] == '\n' && hdr_line) title_end[hidx - 1] = p;     // the header line of record hidx - 1 ends here
            if ((keepmask >> i) & 1u) seq_out[kpos++] = c[i];
        }
        prev = c[i];
    }
}

__global__ void fasta_fill_kernel(unsigned long long* __restrict__ a, uint64_t n, unsigned long long v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

}  // namespace

// ws_fasta layout: totals (256 B) | blk_last_start / carry_ls [nb] | blk_keep [nb] | blk_hdr [nb]
static void fasta_views(po_ctx* ctx, uint64_t nb, fasta_totals** tot, long long** ls, unsigned long long** keep,
                        unsigned long long** hdr) {
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_fasta.p);
    *tot = reinterpret_cast<fasta_totals*>(base);
    *ls = reinterpret_cast<long long*>(base + 256);
    *keep = reinterpret_cast<unsigned long long*>(base + 256 + nb * 8);
    *hdr = reinterpret_cast<unsigned long long*>(base + 256 + 2 * nb * 8);
}

extern "C" int po_fasta_scan_dev(po_ctx* ctx, const uint8_t* d_data, uint64_t len, uint64_t* n_records, uint64_t* seq_bytes) {
    PO_REQUIRE(ctx != nullptr && n_records != nullptr && seq_bytes != nullptr, "po_fasta_scan_dev: NULL argument");
    *n_records = 0;
    *seq_bytes = 0;
    ctx->fasta_data = nullptr;
    if (len == 0) return PO_OK;
    PO_REQUIRE(d_data != nullptr, "po_fasta_scan_dev: NULL buffer");
    PO_REQUIRE((reinterpret_cast<uintptr_t>(d_data) & 15u) == 0, "po_fasta_scan_dev: the file buffer must be 16-byte aligned");
    PO_HIP(hipSetDevice(ctx->device));
    const uint64_t nb = (len + kBlockBytes - 1) / kBlockBytes;
    if (nb >= (1ull << 31)) { po_set_error("po_fasta_scan_dev: file too large for one launch"); return PO_EUNSUPPORTED; }
    int rc = po_buf_reserve(ctx, &ctx->ws_fasta, 256 + 3 * nb * 8);
    if (rc) return rc;
    if (!ctx->h_flag) PO_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_flag), 64, hipHostMallocDefault));
    fasta_totals* tot;
    long long* ls;
    unsigned long long *keep, *hdr;
    fasta_views(ctx, nb, &tot, &ls, &keep, &hdr);
    fasta_totals init;
    memset(&init, 0, sizeof(init));
    init.first_header = len;
    PO_HIP(hipMemcpyAsync(tot, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));                          // `init` is a local
    hipLaunchKernelGGL(fasta_marks_kernel, dim3((uint32_t)nb), dim3(kThreads), 0, ctx->stream, d_data, len, ls, &tot->first_header);
    PO_CHECK_LAUNCH("fasta_marks_kernel");
    hipLaunchKernelGGL(scan_max_kernel, dim3(1), dim3(1024), 0, ctx->stream, ls, nb, ls);      // in place: exclusive carry-ins
    PO_CHECK_LAUNCH("scan_max_kernel");
    hipLaunchKernelGGL(fasta_walk_kernel<false>, dim3((uint32_t)nb), dim3(kThreads), 0, ctx->stream, d_data, len, ls, tot, keep, hdr, tot,
                       nullptr, nullptr, nullptr, nullptr);
    PO_CHECK_LAUNCH("fasta_walk_kernel");
    hipLaunchKernelGGL(scan_sum2_kernel, dim3(1), dim3(1024), 0, ctx->stream, keep, hdr, nb, &tot->seq_bytes, &tot->n_records);
    PO_CHECK_LAUNCH("scan_sum2_kernel");
    static_assert(sizeof(fasta_totals) <= 64, "totals must fit the pinned flag block");
    PO_HIP(hipMemcpyAsync(ctx->h_flag, tot, sizeof(fasta_totals), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    const fasta_totals h = *reinterpret_cast<const fasta_totals*>(ctx->h_flag);
    if (h.junk_before) {
        po_set_error("FASTA input does not start with '>' (%llu non-blank bytes before the first record)", (unsigned long long)h.junk_before);
        return PO_EIO;
    }
    if (h.odd_space) {
        po_set_error("po_fasta_scan_dev: %llu tab / vertical-tab / form-feed bytes on sequence lines: use the host parser (po_fasta_scan)",
                     (unsigned long long)h.odd_space);
        return PO_EUNSUPPORTED;
    }
    *n_records = h.n_records;
    *seq_bytes = h.seq_bytes;
    ctx->fasta_data = d_data;                                          // the partial sums in ws_fasta belong to this buffer
    ctx->fasta_len = len;
    ctx->fasta_records = h.n_records;
    ctx->fasta_seq_bytes = h.seq_bytes;
    return PO_OK;
}

extern "C" int po_fasta_extract_dev(po_ctx* ctx, const uint8_t* d_data, uint64_t len, uint8_t* d_seq, uint64_t* d_offsets,
                                    uint64_t* d_title_begin, uint64_t* d_title_end) {
    PO_REQUIRE(ctx != nullptr, "po_fasta_extract_dev: ctx is NULL");
    if (len == 0) {
        if (d_offsets) PO_HIP(hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), ctx->stream));
        return PO_OK;
    }
    PO_REQUIRE(ctx->fasta_data == d_data && ctx->fasta_len == len,
               "po_fasta_extract_dev: call po_fasta_scan_dev on the same buffer first (it sizes the outputs)");
    PO_REQUIRE(d_offsets && d_title_begin && d_title_end, "po_fasta_extract_dev: NULL buffer");
    PO_HIP(hipSetDevice(ctx->device));
    const uint64_t nb = (len + kBlockBytes - 1) / kBlockBytes;
    fasta_totals* tot;
    long long* ls;
    unsigned long long *keep, *hdr;
    fasta_views(ctx, nb, &tot, &ls, &keep, &hdr);
    fasta_totals h;
    memset(&h, 0, sizeof(h));
    h.n_records = ctx->fasta_records;
    h.seq_bytes = ctx->fasta_seq_bytes;
    PO_REQUIRE(d_seq != nullptr || h.seq_bytes == 0, "po_fasta_extract_dev: NULL sequence buffer");
    unsigned long long* off = reinterpret_cast<unsigned long long*>(d_offsets);
    unsigned long long* tb = reinterpret_cast<unsigned long long*>(d_title_begin);
    unsigned long long* te = reinterpret_cast<unsigned long long*>(d_title_end);
    if (h.n_records) {
        hipLaunchKernelGGL(fasta_fill_kernel, dim3((uint32_t)((h.n_records + 255) / 256)), dim3(256), 0, ctx->stream, te, h.n_records,
                           (unsigned long long)len);                   // a header line the file ends in has no '\n'
        PO_CHECK_LAUNCH("fasta_fill_kernel");
    }
    hipLaunchKernelGGL(fasta_walk_kernel<true>, dim3((uint32_t)nb), dim3(kThreads), 0, ctx->stream, d_data, len, ls, tot, keep, hdr, tot,
                       d_seq, off, tb, te);
    PO_CHECK_LAUNCH("fasta_walk_kernel");
    hipLaunchKernelGGL(fasta_fill_kernel, dim3(1), dim3(1), 0, ctx->stream, off + h.n_records, (uint64_t)1, h.seq_bytes);
    PO_CHECK_LAUNCH("fasta_fill_kernel");
    return PO_OK;
}
