// JSD of near-identical records, evaluated again without cancellation (end of round 5).
//
// The tile kernels compute  JSD = 1/2 (E_a + E_b - S) + ln2/2 (w_a + w_b)  (po_valu_tiles.hip, po_jsd_lut.hip): three sums of
// size ln D whose difference is good to ~1e-14 ABSOLUTE.  The reference's own form
//   phylodist.JSD / KL  (/root/reference/phylopackage/core/phylodist.py:18-24, :43-48):  1/2 sum_k [ x ln(x/m) + y ln(y/m) ],  m = (x + y)/2
// is relatively accurate however small the result is, so records that differ by a handful of k-mers - JSD 1e-8 ... 1e-13 - came out
// 1e-6 ... 1e-2 off in RELATIVE terms (tools/exp/near_dup_metrics.py), and exact duplicates at ~1e-17 instead of 0.
//
// A wave of a JSD tile kernel whose 16 rows of a tile hold a value below 2^-20 appends (tile, first row, rows) to a list
// (po_fix_list; one integer minimum per pair and one ballot per wave in the tile kernels); this kernel then reads those rows of the
// RESULT, and every entry below 2^-20 is evaluated again from the caller's own profiles, word by word, as
//   JSD = 1/2 sum_k m_k phi(t_k),   t = (x - y) / (x + y),   phi(t) = (1 + t) ln(1 + t) + (1 - t) ln(1 - t) = sum_{i>=1} t^(2i) / (i (2i - 1)),
// every term >= 0: the same masking semantics as the reference's KL (a zero numerator contributes nothing: phi(+-1) = 2 ln 2), exactly 0
// for identical frequency vectors, symmetric in the two records.  From integer counts t and m come from exact integer arithmetic
// (t = (a n_b - b n_a) / (a n_b + b n_a)), from frequencies from their float64 difference (exact for nearby values).
#include "po_tiles.h"

namespace {

constexpr int kThreads = 256;
constexpr int kGrid = 512;

__device__ __forceinline__ double phi_term(double t) {
    const double u = t * t;
    if (u <= 0.0625) {                                     // |t| <= 1/4: the series, 14 terms (u^14 / 378 < 1e-19)
        double s = 1.0 / (14.0 * 27.0);
#pragma unroll
        for (int i = 13; i >= 1; --i) s = fma(s, u, 1.0 / ((double)i * (double)(2 * i - 1)));
        return s * u;
    }
    const double p = 1.0 + t, q = 1.0 - t;                 // exact or harmless: no cancellation that matters at |t| > 1/4
    return (p > 0.0 ? p * log(p) : 0.0) + (q > 0.0 ? q * log(q) : 0.0);
}

// counts: [n][dim] uint32 and totals, or freq: [n][dim] float64 (row-major, as the caller passed them; not folded)
__device__ double jsd_exact_pair(const uint32_t* __restrict__ counts, const unsigned long long* __restrict__ totals,
                                 const double* __restrict__ freq, uint32_t dim, uint64_t i, uint64_t j) {
    double acc = 0.0;
    if (counts != nullptr) {
        const unsigned long long na = totals[i], nb = totals[j];
        if (na == 0ull && nb == 0ull) return 0.0;
        const uint32_t* a = counts + i * dim;
        const uint32_t* b = counts + j * dim;
        if (na == 0ull || nb == 0ull) {                    // an empty record against a profile: every word of the profile at t = +-1
            double w = 0.0;
            const uint32_t* c = na ? a : b;
            for (uint32_t k = 0; k < dim; ++k) w += (double)c[k];
            return 0.5 * 0.693147180559945309417232121458 * (w / (double)(na ? na : nb));
        }
        for (uint32_t k = 0; k < dim; ++k) {
            const unsigned long long x = (unsigned long long)a[k] * nb, y = (unsigned long long)b[k] * na;   // < 2^64: counts, totals < 2^32
            if ((x | y) == 0ull) continue;
            const double d = x >= y ? (double)(x - y) : -(double)(y - x);       // the difference is exact before it is rounded
            const double sum = (double)x + (double)y;
            acc = fma(sum, phi_term(d / sum), acc);                              // m_k phi(t_k) 2 n_a n_b
        }
        return 0.25 * acc / ((double)na * (double)nb);
    }
    const double* x = freq + i * dim;
    const double* y = freq + j * dim;
    for (uint32_t k = 0; k < dim; ++k) {
        const double sum = x[k] + y[k];
        if (!(sum > 0.0)) continue;
        acc = fma(sum, phi_term((x[k] - y[k]) / sum), acc);
    }
    return 0.25 * acc;
}

template <typename OUT>
__global__ __launch_bounds__(kThreads) void jsd_exact_rows_kernel(po_tile_args A, const po_fix_list* __restrict__ fix,
                                                                  const uint32_t* __restrict__ counts,
                                                                  const unsigned long long* __restrict__ totals,
                                                                  const double* __restrict__ freq, uint32_t dim) {
    const uint32_t count = fix->count;
    if (count == 0u || count > PO_FIX_CAP) return;         // nothing to do / more than the list holds: all or nothing
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const double below = 0x1p-20;
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const unsigned long long w = fix->entry[e];
        const uint32_t ti = (uint32_t)(w >> 40), tj = (uint32_t)(w >> 16) & 0xFFFFFFu, r0 = (uint32_t)(w >> 8) & 0xFFu, nr = (uint32_t)w & 0xFFu;
        const bool mirrors = po_tile_mirrors(A, ti, tj);
        for (uint32_t p = threadIdx.x; p < nr * 128u; p += kThreads) {
            const uint64_t i = (uint64_t)ti * 128u + r0 + (p >> 7), j = (uint64_t)tj * 128u + (p & 127u);
            if (i < A.row_begin || i >= n_rows || j < A.col_begin || j >= n_cols || i == j) continue;
            OUT* at = out + (i - A.row_begin) * A.ld_out + (j - A.col_begin);
            const double v = (double)*at;
            if (!(v < below)) continue;
            const OUT x = (OUT)jsd_exact_pair(counts, totals, freq, dim, i, j);
            *at = x;
            if (mirrors) mir[(j - A.col_begin) * A.ld_mirror + (i - A.row_begin)] = x;
        }
    }
}

}  // namespace

int po_jsd_exact_reset(po_ctx* ctx, po_fix_list** list) {
    int rc = po_buf_reserve(ctx, &ctx->ws_fix, sizeof(po_fix_list));
    if (rc) return rc;
    *list = static_cast<po_fix_list*>(ctx->ws_fix.p);
    PO_HIP(hipMemsetAsync(ctx->ws_fix.p, 0, 8, ctx->stream));
    return PO_OK;
}

// behind the JSD tile kernels of one block: the rows they noted, from the caller's profiles (counts + totals, or frequencies)
int po_launch_jsd_exact(po_ctx* ctx, const po_tile_args& a, const uint32_t* d_counts, const uint64_t* d_totals, const double* d_freq,
                        uint32_t dim) {
    if (a.fix == nullptr) return PO_OK;
    const unsigned long long* tot = reinterpret_cast<const unsigned long long*>(d_totals);
    if (a.out_f32)
        hipLaunchKernelGGL(jsd_exact_rows_kernel<float>, dim3(kGrid), dim3(kThreads), 0, ctx->stream, a, a.fix, d_counts, tot, d_freq, dim);
    else
        hipLaunchKernelGGL(jsd_exact_rows_kernel<double>, dim3(kGrid), dim3(kThreads), 0, ctx->stream, a, a.fix, d_counts, tot, d_freq, dim);
    PO_CHECK_LAUNCH("jsd_exact_rows_kernel");
    return PO_OK;
}
