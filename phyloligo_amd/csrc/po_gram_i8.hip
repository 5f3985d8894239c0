// Stage 2, Gram-form metrics from EXACT integer dot products on the int8 matrix cores:
// Euclidean distance of count profiles and Spearman's rank correlation.
//
// Same quantities as gram_tile_kernel<Eucl|SC> (po_gram_f64.hip):
//   phylodist.Eucl (/root/reference/phylopackage/core/phylodist.py:36-41)
//     |a-b|^2 = S_a/n_a^2 + S_b/n_b^2 - 2 G/(n_a n_b),   S_x = sum c_x^2,   G = sum c_a c_b   (integer counts c)
//   phylodist.SC (:82-85, as intended: 1 - spearmanr)
//     1 - G / sqrt(N_a N_b),  G = sum r_a r_b,  N_x = sum r_x^2,  r = 2 * centred average rank = 2 #less + #equal - D
// Both G are sums of products of small integers.  A value v is split into 7-bit digits v = 128 hi + lo
// (lo in 0..127, hi signed), each digit plane is an int8 matrix, and
//     G = 16384 <hi_a,hi_b> + 128 (<hi_a,lo_b> + <lo_a,hi_b>) + <lo_a,lo_b>
// comes EXACTLY out of v_mfma_i32_32x32x32_i8 (int32 accumulators, combined in float64 below 2^53).
// One plane covers counts <= 127 (every 2 kb contig at k=4: max 38 measured), two planes cover |v| <= 16383
// (contigs up to ~1 Mb at k=4; ranks for any D <= 16 384), three planes (counts only, round 4) <= 2 097 151: scaffolds and
// chromosomes up to ~250 Mb at k=4 - G = sum over s of 128^s * (sum over p + q = s of <digit_p(a), digit_q(b)>), five int32
// accumulator groups, still exact (G < 2^53 whenever a record's total is below 2^32).  The int8 MFMA rate is ~60x the float64 one, so the
// matrix-core time is small and the kernel is bound by its epilogue and by writing 16 B per pair.
//
// Operand layout (built per call by prep_planes_kernel): plane[p][k/16][record][k%16] - 16-byte K-chunks
// of 128 consecutive records are contiguous (2 KiB), so a tile's operands are staged by LDS-DMA with fully
// coalesced 1 KiB instructions and the MFMA operand reads (16 B per lane, consecutive records) are
// conflict free.  The order of K inside the dot product is irrelevant as long as both operands agree.
//
// Eligibility is decided on the device from the largest |value| of the matrix: every candidate kernel is
// launched and all but one exit at once (no host synchronisation); see po_api.hip.
#include "po_tiles.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128;
constexpr int kThreads = 512;                       // 8 waves: 4 x 2 blocks of 32 x 64 pairs
constexpr int KCH = 128;                            // bytes of K per record and staging step = 4 MFMA k-steps
constexpr int kChunkBytes = 128 * 16;               // one 16-byte K-chunk of the tile's 128 records
constexpr int kTrStride = 33;                       // transposed 32 x 32 block of a wave, in doubles (odd: conflict free)
constexpr int kMirrorBytes = 8 * 32 * kTrStride * 8;
constexpr int kPlanes = 3;                          // digit planes in the workspace (the third one for counts only)
constexpr int kTermBytes = 6 * 128 * 8;             // per-record terms of the tile's rows and columns, read by the epilogue

// planes[p][q][r][16]: digit p of words 16q..16q+15 of record r;  rs: per-record terms;  *maxabs = max |v|.
// A workgroup takes 16 records: thread (rr = t & 15, cl = t >> 4) packs the K-chunks cl, cl+16, .. of record rr,
// so that the 16 threads of one chunk write 256 contiguous bytes.
//   Eucl (SIGNED = false): v = counts;  rs[0][r] = S_r / n_r^2 (S exact), rs[1][r] = 1/n_r (0 for an empty record),
//                                       rs[2][r] = S_r itself (an exact integer below 2^53)
//   SC   (SIGNED = true):  v = r2;      rs[0][r] = N_r = sum r2^2 (exact)
template <bool SIGNED>
__global__ __launch_bounds__(256) void prep_planes_kernel(const uint32_t* __restrict__ vals,
                                                          const unsigned long long* __restrict__ totals, uint64_t n,
                                                          uint32_t dim, uint32_t dpad, uint64_t npad,
                                                          int8_t* __restrict__ planes, double* __restrict__ rs,
                                                          uint32_t* __restrict__ maxabs) {
    __shared__ unsigned long long sq_s[16];
    __shared__ uint32_t mx_s[16];
    const uint32_t t = threadIdx.x, rr = t & 15, cl = t >> 4;
    const uint64_t r = (uint64_t)blockIdx.x * 16 + rr;
    if (t < 16) { sq_s[t] = 0; mx_s[t] = 0; }
    __syncthreads();
    unsigned long long sq = 0;
    uint32_t mx = 0;
    const size_t plane = (size_t)npad * dpad;
    const bool vec = (dim & 3u) == 0;
    for (uint32_t q = cl; q < dpad / 16; q += 16) {
        uint32_t v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0;
        if (r < n) {
            const uint32_t d0 = q * 16;
            if (vec && d0 + 16 <= dim) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 x = *reinterpret_cast<const uint4*>(vals + r * dim + d0 + 4 * g);
                    v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (d0 + e < dim) v[e] = vals[r * dim + d0 + e];
            }
        }
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, top[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int sv = (int)v[e];
            const uint32_t a = SIGNED ? (uint32_t)(sv < 0 ? -sv : sv) : v[e];
            sq += (unsigned long long)a * a;
            mx = max(mx, a);
            lo[e >> 2] |= (v[e] & 127u) << (8 * (e & 3));
            hi[e >> 2] |= (SIGNED ? ((uint32_t)(sv >> 7) & 255u) : ((v[e] >> 7) & 127u)) << (8 * (e & 3));
            if (!SIGNED) top[e >> 2] |= ((v[e] >> 14) & 127u) << (8 * (e & 3));     // third digit of a count (values < 2^21)
        }
        *reinterpret_cast<uint4*>(planes + ((size_t)q * npad + r) * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(planes + plane + ((size_t)q * npad + r) * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        if (!SIGNED) *reinterpret_cast<uint4*>(planes + 2 * plane + ((size_t)q * npad + r) * 16) = make_uint4(top[0], top[1], top[2], top[3]);
    }
    if (sq) atomicAdd(&sq_s[rr], sq);
    if (mx) atomicMax(&mx_s[rr], mx);
    __syncthreads();
    if (cl == 0) {
        if (SIGNED) {
            rs[r] = (double)sq_s[rr];
        } else {
            const unsigned long long tot = (r < n) ? totals[r] : 0ull;
            const double inv = tot ? 1.0 / (double)tot : 0.0;
            rs[r] = (double)sq_s[rr] * (inv * inv);
            rs[npad + r] = inv;
            rs[2 * npad + r] = (double)sq_s[rr];
        }
    }
    if (t == 0) {
        uint32_t m = 0;
        for (int i = 0; i < 16; ++i) m = max(m, mx_s[i]);
        if (m > *maxabs) atomicMax(maxabs, m);            // racy pre-check only skips redundant atomics
        uint32_t* blk = maxabs + 1 + (blockIdx.x >> 3);   // largest |value| of this workgroup's block of 128 records
        if (m > *blk) atomicMax(blk, m);
    }
}

// P digit planes; runs iff  run_above < *maxabs <= run_upto  (maxabs == nullptr: always).
template <int P, int METRIC, typename OUT>
__global__ __launch_bounds__(kThreads, P == 1 ? 4 : 2) void gram_i8_tile_kernel(po_tile_args A, const int8_t* __restrict__ planes,
                                                                               uint32_t dpad, const double* __restrict__ rs,
                                                                               const uint32_t* __restrict__ maxabs,
                                                                               long long run_above, long long run_upto) {
    if (maxabs != nullptr) {
        const long long m = *maxabs;
        if (m <= run_above || m > run_upto) return;
    }
    extern __shared__ __align__(16) unsigned char smem[];  // staging [P][A|B][8 chunks][128 records][16 B], then the mirror scratch
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));   // the wave index as a scalar
    const uint32_t wr = wave >> 1, wc = wave & 1;          // 4 x 2 waves of 32 x 64
    const uint32_t lr = lane & 31, lh = lane >> 5;
    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;
    const size_t plane = (size_t)A.npad * dpad;
    constexpr int kStepUnroll = P == 3 ? 1 : KCH / 32;
    // Three planes: which of them a tile needs is decided by ITS two blocks of records (largest count of each, written by the
    // prep kernel behind *maxabs) - in a real assembly a handful of long contigs need the third digit, most blocks need one -
    // and the planes a block does not need are neither staged nor multiplied (wave-uniform branches; their accumulators stay 0).
    uint32_t pr = P, pc = P;
    if (P == 3 && maxabs != nullptr) {
        const uint32_t mr = maxabs[1 + ti], mc = maxabs[1 + tj];
        pr = mr <= 127u ? 1u : (mr <= 16383u ? 2u : 3u);
        pc = mc <= 127u ? 1u : (mc <= 16383u ? 2u : 3u);
    }
    constexpr int NG = 2 * P - 1;                          // g[s]: sum over p + q = s of <digit p of the row, digit q of the column>

    v16i g[NG][2];
#pragma unroll
    for (int s = 0; s < NG; ++s)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[s][nn][e] = 0;

    for (uint32_t k0 = 0; k0 < dpad; k0 += KCH) {
        __syncthreads();                                   // the previous step's operand reads are done
#pragma unroll
        for (int u = 0; u < 4 * P; ++u) {                  // 32 P one-KiB LDS-DMA instructions, 4 P per wave
            const uint32_t idx = wave * 4 * P + u;
            const uint32_t p = idx >> 5, side = (idx >> 4) & 1, q = (idx >> 1) & 7, half = idx & 1;
            if (P == 3 && p >= (side ? pc : pr)) continue;  // a digit plane that is all zero for this block of records
            const uint64_t rec = (side ? j0 : i0) + half * 64 + lane;
            const int8_t* src = planes + p * plane + ((size_t)(k0 / 16 + q) * A.npad + rec) * 16;
            po_glds16(src, smem + ((p * 2 + side) * 8 + q) * kChunkBytes + half * 1024);
        }
        __syncthreads();                                   // drains the LDS-DMA (vmcnt) of every wave
        // (three planes: 160 accumulator registers of the 256 - the k-steps stay a loop, or the operand fragments of all four
        //  are fetched ahead and the kernel spills)
#pragma unroll kStepUnroll
        for (int s = 0; s < KCH / 32; ++s) {
            const uint32_t q = 2 * s + lh;                 // lane halves take the two 16-byte chunks of a k-step
            v4i a[P], b[P][2];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (P != 3 || (uint32_t)p < pr)
                    a[p] = *reinterpret_cast<const v4i*>(smem + ((p * 2 + 0) * 8 + q) * kChunkBytes + (wr * 32 + lr) * 16);
                if (P != 3 || (uint32_t)p < pc) {
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        b[p][nn] = *reinterpret_cast<const v4i*>(smem + ((p * 2 + 1) * 8 + q) * kChunkBytes + (wc * 64 + nn * 32 + lr) * 16);
                }
            }
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int pa = 0; pa < P; ++pa)
#pragma unroll
                    for (int pb = 0; pb < P; ++pb)
                        if (P != 3 || ((uint32_t)pa < pr && (uint32_t)pb < pc))
                            g[pa + pb][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[pa], b[pb][nn], g[pa + pb][nn], 0, 0, 0);
        }
    }
    __syncthreads();                                       // the staging area becomes the mirror scratch
    // per-record terms into LDS: on gfx9 loads and stores share one in-order counter (vmcnt), so a global load
    // issued among the output stores could only be waited for together with every store before it
    double* terms = reinterpret_cast<double*>(smem + kMirrorBytes);      // [t0 rows | t0 cols | t1 rows | t1 cols]
    if (t < 256) {
        const uint64_t rec = (t < 128) ? i0 + t : j0 + (t - 128);
        terms[t] = rs[rec];
        terms[256 + t] = METRIC == PO_EUCL ? rs[A.npad + rec] : 0.0;
        terms[512 + t] = METRIC == PO_EUCL ? rs[2 * A.npad + rec] : 0.0;
    }
    __syncthreads();

    // ---- epilogue: accumulator layout (32x32): column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const bool mirror = po_tile_mirrors(A, ti, tj);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const uint64_t iw = i0 + wr * 32, jw = j0 + wc * 64;
    double* wl = reinterpret_cast<double*>(smem) + wave * (32 * kTrStride);
    const double* t0r = terms + wr * 32, *t0c = terms + 128 + wc * 64;          // Eucl: S/n^2      SC: N
    const double* t1r = terms + 256 + wr * 32, *t1c = terms + 384 + wc * 64;    // Eucl: 1/n
    const double* t2r = terms + 512 + wr * 32, *t2c = terms + 640 + wc * 64;    // Eucl: S
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const uint64_t c = jw + nn * 32 + lr;
        const double tc = t0c[nn * 32 + lr], ic = METRIC == PO_EUCL ? t1c[nn * 32 + lr] : 0.0;
        const bool c_ok = c >= A.col_begin && c < n_cols;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const uint64_t r = iw + rl;
            const double trr = t0r[rl], irr = METRIC == PO_EUCL ? t1r[rl] : 0.0;
            double G = (double)g[NG - 1][nn][reg];               // Horner in 128: every partial sum an exact integer below 2^53
#pragma unroll
            for (int s = NG - 2; s >= 0; --s) G = fma(128.0, G, (double)g[s][nn][reg]);
            double v;
            if (METRIC == PO_EUCL) {
                const double cross = G * (irr * ic);       // symmetric in (r,c); equals S/n^2 for duplicates
                double d2 = fmax((trr + tc) - 2.0 * cross, 0.0);
                if (d2 <= 1.0e-13 * (trr + tc)) {
                    // cancellation level: are the two count vectors proportional, i.e. the frequency vectors identical
                    // (distance exactly 0 in the reference)?  Cauchy-Schwarz equality G^2 == S_r S_c, tested exactly
                    // with error-free products (all three are integers below 2^53).
                    const double sr = t2r[rl], sc = t2c[nn * 32 + lr];
                    const double p = G * G, pe = fma(G, G, -p), q = sr * sc, qe = fma(sr, sc, -q);
                    if (p == q && pe == qe) d2 = 0.0;
                }
                v = po_sqrt_nonneg(d2);
                if (r == c) v = 0.0;
            } else {                                       // SC; a constant record has N = 0 -> NaN as SciPy gives
                // G / sqrt(N_r N_c) as G * rsqrt: v_rsq_f64 seed + two Newton steps (~1 ulp) instead of the
                // ~40-instruction sqrt + divide; identical records (G = N_r = N_c, exact integers) give exactly 0
                const double x = trr * tc;
                double y = __builtin_amdgcn_rsq(x);
                y = fma(0.5 * y, fma(-(x * y), y, 1.0), y);
                y = fma(0.5 * y, fma(-(x * y), y, 1.0), y);
                v = (G == trr && G == tc) ? (x > 0.0 ? 0.0 : G / x) : 1.0 - G * y;
            }
            if (c_ok && r >= A.row_begin && r < n_rows) po_out_store(&out[(r - A.row_begin) * A.ld_out + (c - A.col_begin)], (OUT)v);
            if (mirror) wl[lr * kTrStride + rl] = v;
        }
        if (mirror) {                                      // wave-private scratch; LDS operations of a wave run in order
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const uint32_t jr = it * 2 + lh;
                const double w = wl[jr * kTrStride + lr];
                const uint64_t cm = jw + nn * 32 + jr, r = iw + lr;
                if (cm >= A.col_begin && cm < n_cols && r >= A.row_begin && r < n_rows)
                    po_out_store(&mir[(cm - A.col_begin) * A.ld_mirror + (r - A.row_begin)], (OUT)w);
            }
        }
    }
}

template <int P, int METRIC>
int launch_tiles(po_ctx* ctx, const po_tile_args& a, const int8_t* planes, uint32_t dpad, const double* rs,
                 const uint32_t* maxabs, long long run_above, long long run_upto, uint64_t* tiles) {
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t staging = (size_t)P * 2 * 8 * kChunkBytes;
    const size_t shmem = (staging > (size_t)kMirrorBytes ? staging : (size_t)kMirrorBytes) + kTermBytes;
    if (a.out_f32) {
        auto k = gram_i8_tile_kernel<P, METRIC, float>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto);
    } else {
        auto k = gram_i8_tile_kernel<P, METRIC, double>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto);
    }
    PO_CHECK_LAUNCH("gram_i8_tile_kernel");
    return PO_OK;
}

struct ws_view {
    int8_t* planes;
    double* rs;
    uint32_t* maxabs;
    uint32_t dpad;
};

ws_view view(void* ws, uint64_t npad, uint32_t dim) {
    ws_view v;
    v.dpad = (uint32_t)po_round_up(dim, KCH);
    uint8_t* base = static_cast<uint8_t*>(ws);
    v.planes = reinterpret_cast<int8_t*>(base);
    v.rs = reinterpret_cast<double*>(base + kPlanes * (size_t)npad * v.dpad);
    v.maxabs = reinterpret_cast<uint32_t*>(base + kPlanes * (size_t)npad * v.dpad + 3 * npad * sizeof(double));
    return v;
}

}  // namespace

// largest count the digit planes represent: three (2^21 - 1) where the int32 accumulators hold them - the middle group adds
// three products <= 127 * 127 per word, 3 * 16129 * 32768 < 2^31 - else two (2 * 16129 * 65536 < 2^31).
uint32_t po_gram_i8_value_limit(uint32_t dim) { return dim <= 32768 ? 2097151u : (dim <= 65536 ? 16383u : 127u); }
bool po_gram_i8_sc_supported(uint32_t dim) { return dim >= 1 && dim <= 16384; }  // |2 #less + #equal - D| <= D - 1 <= 16383: two digits, the high one signed

size_t po_gram_i8_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return kPlanes * npad * po_round_up(dim, KCH) + 3 * npad * sizeof(double) + 256 + (npad / 128) * sizeof(uint32_t);
}

// ws layout: plane lo | plane hi | plane top | rs[3][npad] | maxabs | largest |value| of every block of 128 records.   signed_values: vals are int32 (SC's r2), else uint32 counts.
int po_launch_gram_i8_prep(po_ctx* ctx, const uint32_t* d_vals, const uint64_t* d_totals, bool signed_values, uint64_t n,
                           uint32_t dim, uint64_t npad, void* ws, const uint32_t** maxabs_out) {
    const ws_view v = view(ws, npad, dim);
    PO_HIP(hipMemsetAsync(v.maxabs, 0, (1 + npad / 128) * sizeof(uint32_t), ctx->stream));
    const dim3 grid((uint32_t)(npad / 16));
    if (signed_values)
        hipLaunchKernelGGL(prep_planes_kernel<true>, grid, dim3(256), 0, ctx->stream, d_vals, nullptr, n, dim, v.dpad, npad,
                           v.planes, v.rs, v.maxabs);
    else
        hipLaunchKernelGGL(prep_planes_kernel<false>, grid, dim3(256), 0, ctx->stream, d_vals,
                           reinterpret_cast<const unsigned long long*>(d_totals), n, dim, v.dpad, npad, v.planes, v.rs, v.maxabs);
    PO_CHECK_LAUNCH("prep_planes_kernel");
    if (maxabs_out) *maxabs_out = v.maxabs;
    return PO_OK;
}

// Eucl: the one-, two- and three-plane kernels (max <= 127, <= 16 383, <= limit) are all launched and one of them runs;
// SC: two planes, unconditionally.
int po_launch_gram_i8_tiles(po_ctx* ctx, int metric, const po_tile_args& a, const void* ws, uint64_t* tiles) {
    const ws_view v = view(const_cast<void*>(ws), a.npad, a.dim);
    if (metric == PO_SC) return launch_tiles<2, PO_SC>(ctx, a, v.planes, v.dpad, v.rs, nullptr, 0, 0, tiles);
    if (metric != PO_EUCL) { po_set_error("po_launch_gram_i8_tiles: metric %d is not a Gram-form metric", metric); return PO_EINVAL; }
    int rc = launch_tiles<1, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, -1, 127, tiles);
    if (rc) return rc;
    const uint32_t limit = po_gram_i8_value_limit(a.dim);
    if (limit > 127) rc = launch_tiles<2, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, 127, limit < 16383u ? limit : 16383u, nullptr);
    if (rc) return rc;
    if (limit > 16383) rc = launch_tiles<3, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, 16383, limit, nullptr);
    return rc;
}
