// counts -> frequencies (count2freq, /root/reference/phylopackage/bin/phyloligo.py:633-661) in the
// layouts stage 2 reads, plus the per-row terms the tile kernels fold into their epilogues.
//
// Working layout of stage 2:  Ft[d][npad] = counts[n][d] / totals[n]  (float64, transposed so a
// tile's rows are contiguous along n; columns n..npad-1 and rows dim..roundup(dim,8)-1 are zero).  The division is the same
// correctly-rounded float64 division the reference performs (int/int true division, :656).
#include "po_internal.h"

namespace {

__global__ __launch_bounds__(256) void prep_transpose_kernel(const uint32_t* __restrict__ counts,
                                                             const unsigned long long* __restrict__ totals,
                                                             uint64_t n, uint32_t dim, uint64_t npad,
                                                             double* __restrict__ ft,
                                                             const uint32_t* __restrict__ skip_flag, uint32_t skip_upto) {
    if (skip_flag != nullptr && *skip_flag <= skip_upto) return;      // the int8 path needs no float64 operand
    __shared__ double tile[64][65];
    const uint64_t n0 = (uint64_t)blockIdx.x * 64;
    const uint32_t d0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (uint32_t r = ty; r < 64; r += 4) {                        // r: record within tile, tx: word
        const uint64_t row = n0 + r;
        double v = 0.0;
        if (row < n && d0 + tx < dim) {
            const unsigned long long tot = totals[row];
            if (tot) v = (double)counts[row * dim + d0 + tx] / (double)tot;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 64; r += 4) {                        // r: word within tile, tx: record
        if (d0 + r < ((dim + 7u) & ~7u) && n0 + tx < npad) ft[(uint64_t)(d0 + r) * npad + n0 + tx] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void prep_transpose_freq_kernel(const double* __restrict__ freq, uint64_t n,
                                                                  uint32_t dim, uint64_t npad, double* __restrict__ ft) {
    __shared__ double tile[64][65];
    const uint64_t n0 = (uint64_t)blockIdx.x * 64;
    const uint32_t d0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint64_t row = n0 + r;
        tile[r][tx] = (row < n && d0 + tx < dim) ? freq[row * dim + d0 + tx] : 0.0;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 64; r += 4) {
        if (d0 + r < ((dim + 7u) & ~7u) && n0 + tx < npad) ft[(uint64_t)(d0 + r) * npad + n0 + tx] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void freq_rowmajor_kernel(const uint32_t* __restrict__ counts,
                                                            const unsigned long long* __restrict__ totals,
                                                            uint64_t n, uint32_t dim, double* __restrict__ freq) {
    const uint64_t total = n * (uint64_t)dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long tot = totals[i / dim];
        freq[i] = tot ? (double)counts[i] / (double)tot : 0.0;
    }
}

// rowstat[0][r] = sum_w f ln f (f > 0 terms, library log), rowstat[1][r] = sum_w f (1 up to rounding, 0 for
// an empty record), rowstat[2][r] = sum_w f ln f again but with the SAME table logarithm, instruction for
// instruction, that valu_tile_kernel<JSD> applies to a+b: its (tiny, systematic) errors then cancel in
// 1/2 (E_a + E_b - S), and two identical records come out at rounding level instead of ~1e-13.
// One lane per record, coalesced along n thanks to the transposed layout; fixed summation order (with the
// same doubling point as the tile kernels when the operands are reverse-complement folded, po_fold.hip).
__global__ __launch_bounds__(256) void rowstat_kernel(const double* __restrict__ ft, uint64_t n, uint32_t dim,
                                                      uint64_t npad, double* __restrict__ rowstat,
                                                      const double2* __restrict__ logtab, uint32_t dbl_at) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npad) return;
    const double LN2 = 0.693147180559945309417232121458;
    double e = 0.0, s = 0.0, et = 0.0;
    if (r < n) {
        for (uint32_t d = 0; d < dim; ++d) {
            if (d == dbl_at) { e *= 2.0; s *= 2.0; et *= 2.0; }       // folded operands: two-word orbits end here
            const double f = ft[(uint64_t)d * npad + r];
            if (f > 0.0) {
                e += f * log(f);
                if (logtab != nullptr) {
                    const uint32_t hi = (uint32_t)__double2hiint(f);
                    const double2 te = logtab[((hi >> 11) & 511u) * 4];         // copy 0 of the interval's entry
                    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(f));
                    const double ef = (double)((hi >> 20) & 0x7FFu);
                    const double rr = fma(m, te.x, -1.0);
                    double q = fma(rr, 1.0 / 3.0, -0.5);
                    q = fma(rr, q, 1.0);
                    const double big = fma(ef, LN2, te.y);
                    et = fma(f, fma(rr, q, big), et);
                }
            }
            s += f;
        }
        if (dbl_at != PO_NO_DOUBLING && dbl_at >= dim) { e *= 2.0; s *= 2.0; et *= 2.0; }
    }
    rowstat[r] = e;
    rowstat[npad + r] = s;
    if (logtab != nullptr) rowstat[2 * npad + r] = et;
}

// The same per-record terms straight from integer counts, for calls in which the equal-total kernels own every tile
// (the float64 operand matrix is then never read, so it is not built):
//   rowstat[1][r] = sum_w c / n            rowstat[0][r] = sum_w (c/n) ln (c/n) = (sum_w c ln c) / n - (sum_w c / n) ln n
// One wave per record; folded records (po_fold.hip) count the words before dbl_at twice.
__global__ __launch_bounds__(256) void rowstat_counts_kernel(const uint32_t* __restrict__ counts,
                                                             const unsigned long long* __restrict__ totals, uint64_t n,
                                                             uint32_t dim, uint64_t npad, uint32_t dbl_at, int want_entropy,
                                                             double* __restrict__ rowstat) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= npad) return;
    unsigned long long sum = 0;
    double tsum = 0.0;
    if (r < n) {
        for (uint32_t d = lane; d < dim; d += 64) {
            const uint32_t c = counts[r * dim + d];
            const uint32_t wgt = (dbl_at != PO_NO_DOUBLING && d < dbl_at) ? 2u : 1u;
            sum += (unsigned long long)c * wgt;
            if (want_entropy && c > 1u) tsum += (double)wgt * ((double)c * log((double)c));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_down(sum, o, 64);
        tsum += __shfl_down(tsum, o, 64);
    }
    if (lane == 0) {
        const unsigned long long tot = (r < n) ? totals[r] : 0ull;
        const double w = tot ? (double)sum / (double)tot : 0.0;
        rowstat[npad + r] = w;
        if (want_entropy) rowstat[r] = tot ? tsum / (double)tot - w * log((double)tot) : 0.0;
    }
}

}  // namespace

int po_launch_rowstat_counts(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                             uint64_t npad, double* d_rowstat, bool want_entropy, uint32_t dbl_at) {
    hipLaunchKernelGGL(rowstat_counts_kernel, dim3((uint32_t)((npad + 3) / 4)), dim3(256), 0, ctx->stream, d_counts,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, dim, npad, dbl_at, want_entropy ? 1 : 0, d_rowstat);
    PO_CHECK_LAUNCH("rowstat_counts_kernel");
    return PO_OK;
}

int po_launch_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                   uint64_t npad, double* d_ft, const uint32_t* skip_flag, uint32_t skip_upto) {
    dim3 grid((uint32_t)(npad / 64), (dim + 63) / 64);
    hipLaunchKernelGGL(prep_transpose_kernel, grid, dim3(256), 0, ctx->stream, d_counts,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, dim, npad, d_ft, skip_flag, skip_upto);
    PO_CHECK_LAUNCH("prep_transpose_kernel");
    return PO_OK;
}

int po_launch_prep_freq(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, uint64_t npad, double* d_ft) {
    dim3 grid((uint32_t)(npad / 64), (dim + 63) / 64);
    hipLaunchKernelGGL(prep_transpose_freq_kernel, grid, dim3(256), 0, ctx->stream, d_freq, n, dim, npad, d_ft);
    PO_CHECK_LAUNCH("prep_transpose_freq_kernel");
    return PO_OK;
}

int po_launch_freq_rowmajor(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n,
                            uint32_t dim, double* d_freq) {
    if (n == 0) return PO_OK;
    const uint64_t total = n * (uint64_t)dim;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(freq_rowmajor_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_counts,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, dim, d_freq);
    PO_CHECK_LAUNCH("freq_rowmajor_kernel");
    return PO_OK;
}

int po_launch_rowstat(po_ctx* ctx, const double* d_ft, uint64_t n, uint32_t dim, uint64_t npad, double* d_rowstat,
                      const void* logtab, uint32_t dbl_at) {
    hipLaunchKernelGGL(rowstat_kernel, dim3((uint32_t)((npad + 255) / 256)), dim3(256), 0, ctx->stream, d_ft, n, dim,
                       npad, d_rowstat, static_cast<const double2*>(logtab), dbl_at);
    PO_CHECK_LAUNCH("rowstat_kernel");
    return PO_OK;
}
