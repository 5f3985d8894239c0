// Stage 1: exact integer spaced-word profiles of every record, straight from the sequence bytes.
//
// Replaces the per-record Python hot loop of the reference
//   select_strand -> upper() -> cut_sequence_and_count_pattern -> count2freq ordering
//   (/root/reference/phylopackage/bin/phyloligo.py:124-149, :683, :601-631, :653)
// with one byte scan: HBM-bound (1 byte per base read once, 4*4^k bytes per record written).
//
// Work decomposition: a record of L bases is cut into ceil(L/SPAN) chunks, one workgroup per
// chunk (records of a 2 kb assembly are one chunk each; a 10 Mb chromosome is 2.4k chunks).
// A workgroup stages its bytes with aligned 16-byte loads, decodes them once to 2-bit digits
// (C=0,G=1,A=2,T=3; anything else breaks a word) in LDS, then every lane slides a 2*W-bit
// rolling register over 16 consecutive window starts.  Both strands come out of the same
// pass: the forward register gives the '+' word, a second register filled from the other
// end with complemented digits (digit XOR 1) gives the word the same window spells on the
// reverse-complement strand.  Words go to a private LDS histogram (ds_add_u32); the W-1
// words spanning the seq|revcomp(seq) junction of `-s both` (phyloligo.py:141) are added by
// the chunk that holds the record's end.
#include "po_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPerLane = 16;                       // window starts per lane
constexpr int kTile = kThreads * kPerLane;         // 4096 staged positions carrying a window start
constexpr int kSpan = kTile - 16;                  // window starts per chunk (16 B alignment slack)
constexpr int kHalo = 48;                          // >= W-1 (31), keeps the staging a multiple of 16 B
constexpr int kStage = kTile + kHalo;              // staged bytes per chunk
constexpr uint32_t kMaxLdsBins = 16384;            // 64 KiB histogram; above that count in HBM directly

struct CountParams {
    uint32_t window, k, dim, nruns, patbits;
    uint32_t src_shift[PO_MAX_RUNS];
    uint32_t dst_shift[PO_MAX_RUNS];
    uint32_t mask[PO_MAX_RUNS];
    int strand;
    uint32_t n_seqs;
    uint64_t total_bytes;
};

// digit of a base, 4 = not A/C/G/T.  (c>>1)&3 is A0 C1 T2 G3 in either case; 0x72 reorders
// that to C0 G1 A2 T3; membership via a 32-bit set over (upper(c) - 'A').
__device__ __forceinline__ uint32_t base_digit(uint32_t c) {
    const uint32_t x = (c & 0xDFu) - 0x41u;
    const uint32_t member = (x < 32u) ? ((0x00080045u >> x) & 1u) : 0u;   // A=0 C=2 G=6 T=19
    const uint32_t d = (0x72u >> (((c >> 1) & 3u) * 2u)) & 3u;
    return member ? d : 4u;
}

__device__ __forceinline__ uint32_t word_index(uint64_t reg, const CountParams& P) {
    uint32_t idx = 0;
    for (uint32_t r = 0; r < P.nruns; ++r)
        idx |= ((uint32_t)(reg >> P.src_shift[r]) & P.mask[r]) << P.dst_shift[r];
    return idx;
}

// chunks per record + exclusive scan -> chunk_start[n+1].  One workgroup; lanes own slices.
__global__ __launch_bounds__(1024) void chunk_scan_kernel(const uint64_t* __restrict__ offsets, uint32_t n,
                                                          uint32_t* __restrict__ chunk_start) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t len = offsets[i + 1] - offsets[i];
        sum += (uint32_t)((len + kSpan - 1) / kSpan);
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {      // Hillis-Steele inclusive scan
        uint32_t v = (t >= d) ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;                  // exclusive prefix of this lane's slice
    for (uint32_t i = lo; i < hi; ++i) {
        chunk_start[i] = run;
        const uint64_t len = offsets[i + 1] - offsets[i];
        run += (uint32_t)((len + kSpan - 1) / kSpan);
    }
    if (t == 1023) chunk_start[n] = part[1023];
}

template <bool LDS_HIST>
__global__ __launch_bounds__(kThreads) void count_kernel(const uint8_t* __restrict__ seq,
                                                         const uint64_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ chunk_start,
                                                         CountParams P, uint32_t* __restrict__ counts,
                                                         unsigned long long* __restrict__ totals) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint8_t* codes = reinterpret_cast<uint8_t*>(smem);            // [kStage]
    uint32_t* hist = smem + kStage / 4;                            // [dim] when LDS_HIST
    uint32_t* blk_total = hist + (LDS_HIST ? P.dim : 0);           // [1]

    const uint32_t b = blockIdx.x;
    const uint32_t nchunks = chunk_start[P.n_seqs];
    if (b >= nchunks) return;                                      // grid is an upper bound
    uint32_t lo = 0, hi = P.n_seqs;                                // last record with chunk_start <= b
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= b) lo = mid; else hi = mid;
    }
    const uint32_t rec = lo;
    const uint32_t chunk = b - chunk_start[rec];
    const uint32_t rec_chunks = chunk_start[rec + 1] - chunk_start[rec];
    const uint64_t off = offsets[rec];
    const int64_t L = (int64_t)(offsets[rec + 1] - off);
    const int64_t p_lo = (int64_t)chunk * kSpan;                   // window starts [p_lo, p_hi) are ours
    const int64_t p_hi = min(p_lo + (int64_t)kSpan, L);
    const uint64_t a0 = (off + (uint64_t)p_lo) & ~(uint64_t)15;    // 16 B aligned staging origin
    const int64_t pos0 = (int64_t)a0 - (int64_t)off;               // record position of staged byte 0

    const uint32_t t = threadIdx.x;
    if (LDS_HIST)
        for (uint32_t d = t; d < P.dim; d += kThreads) hist[d] = 0;
    if (t == 0) *blk_total = 0;

    // ---- stage + decode -------------------------------------------------------------------
    for (uint32_t v = t; v < kStage / 16; v += kThreads) {
        const uint64_t a = a0 + (uint64_t)v * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        if (a + 16 <= P.total_bytes) {
            const uint4 q = *reinterpret_cast<const uint4*>(seq + a);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        } else {
            for (int i = 0; i < 16; ++i)
                if (a + i < P.total_bytes) w[i >> 2] |= (uint32_t)seq[a + i] << (8 * (i & 3));
        }
        const int64_t pos = pos0 + (int64_t)v * 16;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t d = base_digit((w[j] >> (8 * i)) & 0xFFu);
                if (pos + j * 4 + i >= L) d = 4u;                   // bytes of the next record
                packed |= d << (8 * i);
            }
            o[j] = packed;
        }
        *reinterpret_cast<uint4*>(codes + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // ---- slide ------------------------------------------------------------------------------
    const uint32_t W = P.window;
    const bool want_plus = P.strand != PO_STRAND_MINUS;
    const bool want_minus = P.strand != PO_STRAND_PLUS;
    uint32_t mine = 0;
    {
        const uint4 q0 = *reinterpret_cast<const uint4*>(codes + t * 16);
        const uint4 q1 = *reinterpret_cast<const uint4*>(codes + t * 16 + 16);
        const uint4 q2 = *reinterpret_cast<const uint4*>(codes + t * 16 + 32);
        const uint32_t cw[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        const int64_t first = pos0 + (int64_t)t * kPerLane;        // record position of this lane's start 0
        uint64_t fwd = 0, rev = 0;
        uint32_t run = 0;
        const uint32_t top = 2 * W - 2;
#pragma unroll
        for (int i = 0; i < kPerLane + PO_MAX_WINDOW - 1; ++i) {
            if (i < (int)(kPerLane + W - 1)) {                      // uniform
                const uint32_t d = (cw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                run = (d < 4u) ? run + 1u : 0u;
                fwd = (fwd << 2) | (uint64_t)(d & 3u);
                rev = (rev >> 2) | ((uint64_t)((d & 3u) ^ 1u) << top);
                const int s = i - (int)(W - 1);                     // window start index of this lane
                if (s >= 0) {
                    const int64_t p = first + s;
                    if (run >= W && p >= p_lo && p < p_hi) {
                        if (want_plus) {
                            const uint32_t idx = word_index(fwd, P);
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                            ++mine;
                        }
                        if (want_minus) {
                            const uint32_t idx = word_index(rev, P);
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                            ++mine;
                        }
                    }
                }
            }
        }
    }

    // ---- junction words of seq + revcomp(seq) (-s both), by the chunk holding the record end ----
    if (P.strand == PO_STRAND_BOTH && chunk == rec_chunks - 1 && t < W - 1) {
        const int64_t p = L - (int64_t)W + 1 + (int64_t)t;          // start in the 2L-long virtual string
        if (p >= 0 && p < L && p + (int64_t)W <= 2 * L) {
            uint32_t idx = 0;
            bool ok = true;
            for (uint32_t x = 0; x < W; ++x) {
                const int64_t q = p + x;
                const bool fw = q < L;
                uint32_t d = base_digit(seq[off + (uint64_t)(fw ? q : 2 * L - 1 - q)]);
                ok = ok && (d < 4u);
                if (!fw) d ^= 1u;
                if ((P.patbits >> x) & 1u) idx = idx * 4u + (d & 3u);
            }
            if (ok) {
                if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                ++mine;
            }
        }
    }

    // ---- totals: wave reduce, one LDS add per wave, one HBM add per chunk ------------------------
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((t & 63u) == 0 && mine) atomicAdd(blk_total, mine);
    __syncthreads();
    if (t == 0 && *blk_total) atomicAdd(&totals[rec], (unsigned long long)*blk_total);

    // ---- flush ------------------------------------------------------------------------------
    if (LDS_HIST) {
        uint32_t* row = counts + (uint64_t)rec * P.dim;
        if (rec_chunks == 1) {
            for (uint32_t d = t; d < P.dim; d += kThreads) row[d] = hist[d];
        } else {
            for (uint32_t d = t; d < P.dim; d += kThreads) {
                const uint32_t v = hist[d];
                if (v) atomicAdd(&row[d], v);
            }
        }
    }
}

}  // namespace

int po_launch_count(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_offsets, uint64_t n_seqs,
                    uint64_t total_bytes, const po_pattern& pat, int strand, uint32_t* d_counts,
                    uint64_t* d_totals) {
    if (n_seqs == 0) return PO_OK;
    if (n_seqs >= (1ull << 31)) { po_set_error("too many records (%llu)", (unsigned long long)n_seqs); return PO_EUNSUPPORTED; }
    const uint64_t max_chunks = total_bytes / kSpan + n_seqs;
    if (max_chunks >= (1ull << 31)) { po_set_error("input too large for one launch"); return PO_EUNSUPPORTED; }

    int rc = po_buf_reserve(ctx, &ctx->ws_aux, (n_seqs + 1) * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t* chunk_start = static_cast<uint32_t*>(ctx->ws_aux.p);

    PO_HIP(hipMemsetAsync(d_counts, 0, n_seqs * (uint64_t)pat.dim * sizeof(uint32_t), ctx->stream));
    PO_HIP(hipMemsetAsync(d_totals, 0, n_seqs * sizeof(uint64_t), ctx->stream));

    hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_offsets, (uint32_t)n_seqs, chunk_start);
    PO_CHECK_LAUNCH("chunk_scan_kernel");

    CountParams P;
    memset(&P, 0, sizeof(P));
    P.window = pat.window; P.k = pat.k; P.dim = pat.dim; P.nruns = pat.nruns;
    for (uint32_t r = 0; r < pat.nruns; ++r) {
        P.src_shift[r] = pat.src_shift[r]; P.dst_shift[r] = pat.dst_shift[r]; P.mask[r] = pat.mask[r];
    }
    for (uint32_t i = 0; i < pat.k; ++i) P.patbits |= 1u << pat.ones[i];
    P.strand = strand;
    P.n_seqs = (uint32_t)n_seqs;
    P.total_bytes = total_bytes;

    const bool lds_hist = pat.dim <= kMaxLdsBins;
    const size_t shmem = kStage + (lds_hist ? (size_t)pat.dim * 4 : 0) + 16;
    unsigned long long* tot = reinterpret_cast<unsigned long long*>(d_totals);
    if (lds_hist) {
        hipLaunchKernelGGL(count_kernel<true>, dim3((uint32_t)max_chunks), dim3(kThreads), shmem, ctx->stream,
                           d_seq, d_offsets, chunk_start, P, d_counts, tot);
    } else {
        hipLaunchKernelGGL(count_kernel<false>, dim3((uint32_t)max_chunks), dim3(kThreads), shmem, ctx->stream,
                           d_seq, d_offsets, chunk_start, P, d_counts, tot);
    }
    PO_CHECK_LAUNCH("count_kernel");
    return PO_OK;
}
