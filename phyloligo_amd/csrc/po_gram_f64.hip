// Stage 2, Gram-form metrics on the float64 matrix cores: Euclidean distance and Spearman.
//
// Replaces one Python call per contig pair
//   phylodist.Eucl  (/root/reference/phylopackage/core/phylodist.py:36-41)
//   phylodist.SC    (/root/reference/phylopackage/core/phylodist.py:82-85, as intended)
// under sklearn's pairwise_distances (/root/reference/phylopackage/bin/phyloligo.py:364-392).
//
//   Eucl(a,b)^2 = |a|^2 + |b|^2 - 2 a.b            SC(a,b) = 1 - r_a.r_b / sqrt(|r_a|^2 |r_b|^2)
// (r = centred average ranks).  The dot products are a dense X X^T over the transposed
// float64 operand matrix Xt[d][npad]: v_mfma_f64_16x16x4_f64, 128 x 128 tile per workgroup,
// 64 x 64 per wave (4 x 4 MFMA tiles, 64 float64 accumulators per lane), operands staged
// 8 words at a time through double-buffered LDS.  The squared norms come from the SAME
// instruction sequence run on the diagonal 16 x 16 blocks, so that two identical records give
// |a|^2 + |a|^2 - 2 a.a = 0 exactly, as the reference's (a-b)^2 form does.
#include "po_tiles.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128;
constexpr int KC = 16;                            // words staged per step: 4 MFMA k-steps (4096 MFMA cycles) cover the
                                                  // global-load latency of the next step's staging
constexpr int kThreads = 256;
constexpr int kRowStride = TM + TN + 16;          // +16 doubles: k-rows of one MFMA operand land in
constexpr int kStageDoubles = KC * kRowStride;    // different bank halves (ds_read_b64, 32-lane groups)
constexpr int kTrStride = 66;                     // transposed 16 x 64 block of a wave: lc*66 + lg hits 32 distinct bank pairs




// |x_r|^2 for 16 records per wave, by the same k-ascending MFMA chain the tile kernel uses.
__global__ __launch_bounds__(64) void gram_diag_kernel(const double* __restrict__ xt, uint32_t dim, uint64_t npad,
                                                       double* __restrict__ norms, const uint32_t* __restrict__ skip_flag,
                                                       uint32_t skip_upto) {
    if (skip_flag != nullptr && *skip_flag <= skip_upto) return;
    const uint32_t l = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 16;
    const uint32_t c = l & 15, g = l >> 4;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (uint32_t k0 = 0; k0 < dim; k0 += 4) {
        const uint32_t k = k0 + g;
        const double v = (k < dim) ? xt[(uint64_t)k * npad + r0 + c] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
    }
    // lane holds rows g, g+4, g+8, g+12 of column c; the diagonal element of c sits where row == c
    if ((c & 3u) == g) {
        const uint32_t reg = (c - g) >> 2;
        const double d = reg == 0 ? acc[0] : reg == 1 ? acc[1] : reg == 2 ? acc[2] : acc[3];
        norms[r0 + c] = d;
    }
}

template <int METRIC, typename OUT>
__global__ __launch_bounds__(kThreads, 2) void gram_tile_kernel(po_tile_args A, const double* __restrict__ norms,
                                                                const uint32_t* __restrict__ i8flag, uint32_t i8_upto) {
    if (i8flag != nullptr && *i8flag <= i8_upto) return;   // an exact int8 kernel owns the matrix (po_gram_i8.hip)
    extern __shared__ __align__(16) unsigned char smem[];
    double* stage = reinterpret_cast<double*>(smem);                    // [2][KC][kRowStride]

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    const uint32_t wr = wave >> 1, wc = wave & 1;                       // 2 x 2 waves of 64 x 64
    const uint32_t lc = lane & 15, lg = lane >> 4;

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // staging through registers: lane loads 4 consecutive records of words (k0 + sk) and (k0 + sk + 8)
    // for the A block and for the B block (32 B each), issued a whole step ahead of their use
    const uint32_t sk = t >> 5, sc = (t & 31) * 4;
    const double* gA = A.ft + i0 + sc;
    const double* gB = A.ft + j0 + sc;
    double2 ra[2][2], rb[2][2];
    auto gload = [&](uint32_t k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t k = k0 + sk + 8 * h;
            if (k < A.dim) {
                const double* pa = gA + (uint64_t)k * A.npad;
                const double* pb = gB + (uint64_t)k * A.npad;
                ra[h][0] = *reinterpret_cast<const double2*>(pa);
                ra[h][1] = *reinterpret_cast<const double2*>(pa + 2);
                rb[h][0] = *reinterpret_cast<const double2*>(pb);
                rb[h][1] = *reinterpret_cast<const double2*>(pb + 2);
            } else {
                ra[h][0] = ra[h][1] = rb[h][0] = rb[h][1] = make_double2(0.0, 0.0);
            }
        }
    };
    auto sstore = [&](uint32_t buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double* s = stage + buf * kStageDoubles + (sk + 8 * h) * kRowStride;
            *reinterpret_cast<double2*>(s + sc) = ra[h][0];
            *reinterpret_cast<double2*>(s + sc + 2) = ra[h][1];
            *reinterpret_cast<double2*>(s + TM + sc) = rb[h][0];
            *reinterpret_cast<double2*>(s + TM + sc + 2) = rb[h][1];
        }
    };

    gload(0);
    sstore(0);
    __syncthreads();

    uint32_t cur = 0;
    for (uint32_t k0 = 0; k0 < A.dim; k0 += KC) {
        const bool more = k0 + KC < A.dim;
        if (more) gload(k0 + KC);
        const double* s = stage + cur * kStageDoubles;
        // fragments of k-step ks+1 are read from LDS while the 16 MFMAs of k-step ks issue
        double fa[2][4], fb[2][4];
        auto frag = [&](int ks, double (&a)[4], double (&b)[4]) {
            const double* srow = s + (ks * 4 + lg) * kRowStride;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                a[m] = srow[wr * 64 + m * 16 + lc];
                b[m] = srow[TM + wc * 64 + m * 16 + lc];
            }
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            if (ks + 1 < KC / 4) frag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds rows lg + 4*reg, column lc of each 16 x 16 block -------------------
    // The tile itself leaves as 128-byte row pieces (16 lanes x 8 B).  The mirrored tile is transposed
    // through wave-private LDS (the staging buffers are free now), 16 columns at a time, so that it
    // leaves as 512-byte row pieces of 16-byte stores instead of 32-byte crumbs.
    const bool mirror = po_tile_mirrors(A, ti, tj);
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const uint64_t iw = i0 + wr * 64, jw = j0 + wc * 64;
    double* wl = stage + wave * (16 * kTrStride);
    const bool vec_mir = sizeof(OUT) == 8 && (A.ld_mirror & 1) == 0 && iw >= A.row_begin && ((iw - A.row_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.mirror) & 15) == 0;
    double ni[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ni[m][reg] = norms[iw + m * 16 + lg + 4 * reg];
    double njs[4];                                         // all loads before the first store: on gfx9 loads and stores
#pragma unroll                                             // share the in-order vmcnt counter
    for (int n = 0; n < 4; ++n) njs[n] = norms[jw + n * 16 + lc];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const uint64_t j = jw + n * 16 + lc;
        const double nj = njs[n];
        const bool j_ok = j >= A.col_begin && j < n_cols;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const uint64_t i = iw + m * 16 + lg + 4 * reg;
                const double g = acc[m][n][reg];
                double v;
                if (METRIC == PO_EUCL) {
                    const double sum = ni[m][reg] + nj;
                    double d2 = sum - 2.0 * g;
                    // Cancellation level (near-identical records; d2 == 0 included - identical records give it by construction, but so
                    // does a pair whose distance is below 1e-8 of its norms: an adversarial fuzz found 2.5e-9 returned as 0): the
                    // three rounded terms have lost their leading bits to each other - at d^2 = 1e-11 sum the distance was good to
                    // 2e-6 only (a fuzz seed at the end of round 5).  Such a pair is evaluated the way the reference does it, as the
                    // sum of squared differences over the words (operand matrix in HBM; a rare, divergent loop).  From 2^-16 on: the Gram
                    // entry itself carries the rounding of up to D additions here (4 096 words: 4.5e-13 at worst, 1.5e-8 of a distance at the threshold).
                    if (d2 <= 0x1p-16 * sum && i != j && i < A.n && j < A.n) {
                        const double* pi = A.ft + i;
                        const double* pj = A.ft + j;
                        double acc2 = 0.0;
                        for (uint32_t k = 0; k < A.dim; ++k) {
                            const double df = pi[(uint64_t)k * A.npad] - pj[(uint64_t)k * A.npad];
                            acc2 = fma(df, df, acc2);
                        }
                        d2 = acc2;
                    }
                    v = po_sqrt_nonneg(fmax(d2, 0.0));
                    if (i == j) v = 0.0;
                } else {  // PO_SC: 1 - Pearson correlation of the centred ranks; constant row -> NaN
                    v = 1.0 - g / sqrt(ni[m][reg] * nj);
                }
                if (j_ok && i >= A.row_begin && i < n_rows) po_out_store(&out[(i - A.row_begin) * A.ld_out + (j - A.col_begin)], (OUT)v);
                if (mirror) wl[lc * kTrStride + m * 16 + lg + 4 * reg] = v;
            }
        }
        if (mirror) {                                      // wave-uniform; LDS operations of one wave execute in order
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const uint32_t jr = it * 2 + (lane >> 5), ic = 2 * (lane & 31);
                const double2 w = *reinterpret_cast<const double2*>(wl + jr * kTrStride + ic);
                const uint64_t jm = jw + n * 16 + jr, i = iw + ic;
                if (jm < A.col_begin || jm >= n_cols) continue;
                OUT* row = mir + (jm - A.col_begin) * A.ld_mirror;
                if (vec_mir && i + 1 < n_rows) {
                    *reinterpret_cast<double2*>(row + (i - A.row_begin)) = w;
                } else {
                    if (i >= A.row_begin && i < n_rows) po_out_store(&row[i - A.row_begin], (OUT)w.x);
                    if (i + 1 >= A.row_begin && i + 1 < n_rows) po_out_store(&row[i + 1 - A.row_begin], (OUT)w.y);
                }
            }
        }
    }
}

template <int METRIC>
int launch_gram(po_ctx* ctx, const po_tile_args& a, const double* norms, const uint32_t* i8flag, uint32_t i8_upto, uint64_t* tiles) {
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = 2 * kStageDoubles * sizeof(double);
    PO_SHMEM(ctx, (gram_tile_kernel<METRIC, float>), shmem);
    PO_SHMEM(ctx, (gram_tile_kernel<METRIC, double>), shmem);
    if (a.out_f32)
        hipLaunchKernelGGL((gram_tile_kernel<METRIC, float>), dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, norms, i8flag, i8_upto);
    else
        hipLaunchKernelGGL((gram_tile_kernel<METRIC, double>), dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, norms, i8flag, i8_upto);
    PO_CHECK_LAUNCH("gram_tile_kernel");
    return PO_OK;
}

}  // namespace

// squared norms of every operand row into rowstat[2][npad] (once per problem, before the tile launches)
int po_launch_gram_norms(po_ctx* ctx, const double* ft, uint32_t dim, uint64_t npad, double* rowstat, const uint32_t* skip_flag,
                         uint32_t skip_upto) {
    hipLaunchKernelGGL(gram_diag_kernel, dim3((uint32_t)(npad / 16)), dim3(64), 0, ctx->stream, ft, dim, npad, rowstat + 2 * npad, skip_flag, skip_upto);
    PO_CHECK_LAUNCH("gram_diag_kernel");
    return PO_OK;
}

// a.ft is the operand matrix (frequencies for Eucl, centred ranks for SC)
int po_launch_gram_f64(po_ctx* ctx, int metric, const po_tile_args& a, const uint32_t* i8flag, uint32_t i8_upto, uint64_t* tiles) {
    const double* norms = a.rowstat + 2 * a.npad;
    if (metric == PO_EUCL) return launch_gram<PO_EUCL>(ctx, a, norms, i8flag, i8_upto, tiles);
    if (metric == PO_SC) return launch_gram<PO_SC>(ctx, a, norms, nullptr, 0, tiles);
    po_set_error("po_launch_gram_f64: metric %d is not a Gram-form metric", metric);
    return PO_EINVAL;
}
