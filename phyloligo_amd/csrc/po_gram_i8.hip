// Stage 2, Euclidean distance from EXACT integer dot products on the int8 matrix cores.
//
// Same quantity as gram_tile_kernel<Eucl> (phylodist.Eucl,
// /root/reference/phylopackage/core/phylodist.py:36-41):
//     |a-b|^2 = S_a/n_a^2 + S_b/n_b^2 - 2 G/(n_a n_b),   S_x = sum c_x^2,   G = sum c_a c_b
// When no count exceeds 127 (every 2 kb contig at k=4; max 38 measured) the profiles fit int8 and
// v_mfma_i32_32x32x32_i8 gives G exactly, at ~60x the float64 MFMA rate: the matrix-core time becomes
// negligible and the kernel is bound by writing the 16 B per pair of output.  Because the MFMA work is
// nearly free, the transposed tile is produced by a second MFMA with the operands swapped (G^T = B A^T)
// instead of a transposing store: both the tile and its mirror are written as full 256-byte row
// segments straight from the accumulator layout (lanes 0..31 = 32 consecutive columns).
//
// Operands come straight from the L2 / Infinity-Cache resident int8 matrix (12.8 MB at N=50k, D=256);
// no LDS staging is needed.  Eligibility (max count <= 127) is decided on the device: this kernel exits
// when the flag is clear, gram_tile_kernel<Eucl> exits when it is set (po_gram_f64.hip).
#include "po_tiles.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128;
constexpr int kThreads = 256;

// c8[r][dpad] = (int8)counts[r][d], zero padded (rows n..npad-1, words dim..dpad-1);
// rs[0][r] = S_r * (1/n_r * 1/n_r) with S_r = sum c^2 (exact integer), rs[1][r] = 1/n_r (0 for an empty
// record); *maxcount = max over the matrix.  One wave per record, 4 words per lane and step.
__global__ __launch_bounds__(256) void prep_i8_kernel(const uint32_t* __restrict__ counts,
                                                      const unsigned long long* __restrict__ totals, uint64_t n,
                                                      uint32_t dim, uint32_t dpad, int8_t* __restrict__ c8,
                                                      double* __restrict__ rs, uint64_t npad,
                                                      uint32_t* __restrict__ maxcount) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= npad) return;
    unsigned long long sq = 0;
    uint32_t mx = 0;
    const bool vec = (dim & 3u) == 0;
    for (uint32_t d = lane * 4; d < dpad; d += 256) {
        uint32_t v[4] = {0, 0, 0, 0};
        if (r < n) {
            if (vec && d + 4 <= dim) {
                const uint4 q = *reinterpret_cast<const uint4*>(counts + r * dim + d);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (d + e < dim) v[e] = counts[r * dim + d + e];
            }
        }
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sq += (unsigned long long)v[e] * v[e];
            mx = max(mx, v[e]);
            packed |= min(v[e], 127u) << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(c8 + r * dpad + d) = packed;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_down(sq, o, 64);
        mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
    }
    if (lane == 0) {
        const unsigned long long tot = (r < n) ? totals[r] : 0ull;
        const double inv = tot ? 1.0 / (double)tot : 0.0;
        rs[r] = (double)sq * (inv * inv);
        rs[npad + r] = inv;
        if (mx > *maxcount) atomicMax(maxcount, mx);      // racy pre-check only skips redundant atomics
    }
}

// One orientation of a wave's 64 x 64 block: rows = records r0.., columns = records c0...  Values go to
// dst[(row - row_off) * ld + (col - col_off)]; `swap` says that rows are the block's columns (mirror).
template <typename OUT>
__device__ __forceinline__ void gram_i8_half(const po_tile_args& A, const int8_t* __restrict__ c8, uint32_t dpad,
                                             const double* __restrict__ T, const double* __restrict__ inv,
                                             uint64_t r0, uint64_t c0, bool swap, uint32_t lr, uint32_t lh) {
    v16i g[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[m][q][e] = 0;
    // lane holds 16 consecutive words of record (lr) starting at word 16*lh of the K step
    const int8_t* pa = c8 + (r0 + lr) * dpad + 16 * lh;
    const int8_t* pb = c8 + (c0 + lr) * dpad + 16 * lh;
    for (uint32_t k0 = 0; k0 < dpad; k0 += 32) {
        v4i a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[m] = *reinterpret_cast<const v4i*>(pa + (uint64_t)m * 32 * dpad + k0);
            b[m] = *reinterpret_cast<const v4i*>(pb + (uint64_t)m * 32 * dpad + k0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) g[m][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[q], g[m][q], 0, 0, 0);
    }
    // accumulator layout (32x32): column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    OUT* dst = static_cast<OUT*>(swap ? A.mirror : A.out);
    const uint64_t ld = swap ? A.ld_mirror : A.ld_out;
    const uint64_t row_off = swap ? A.col_begin : A.row_begin, col_off = swap ? A.row_begin : A.col_begin;
    const uint64_t row_hi = min(A.n, swap ? A.col_end : A.row_end), col_hi = min(A.n, swap ? A.row_end : A.col_end);
    // wave-uniform: is the whole 64 x 64 block inside the output block, and can it touch the diagonal?
    const bool interior = r0 >= row_off && r0 + 64 <= row_hi && c0 >= col_off && c0 + 64 <= col_hi;
    const bool on_diag = r0 == c0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        double tr[16], ir[16];                            // row terms of this lane's 16 rows
        const uint64_t rbase = r0 + m * 32 + 4 * lh;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const uint64_t r = rbase + (reg & 3) + 8 * (reg >> 2);
            tr[reg] = T[r];
            ir[reg] = inv[r];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint64_t c = c0 + q * 32 + lr;
            const double tc = T[c], ic = inv[c];
            OUT* col = dst + (c - col_off);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint64_t r = rbase + (reg & 3) + 8 * (reg >> 2);
                const double cross = (double)g[m][q][reg] * (ir[reg] * ic);   // symmetric in (r,c); = T for duplicates
                double v = po_sqrt_nonneg(fmax((tr[reg] + tc) - 2.0 * cross, 0.0));
                if (on_diag && r == c) v = 0.0;
                if (interior || (r >= row_off && r < row_hi && c >= col_off && c < col_hi)) col[(r - row_off) * ld] = (OUT)v;
            }
        }
    }
}

template <typename OUT>
__global__ __launch_bounds__(kThreads, 2) void gram_i8_tile_kernel(po_tile_args A, const int8_t* __restrict__ c8,
                                                                   uint32_t dpad, const double* __restrict__ rs,
                                                                   const uint32_t* __restrict__ maxcount) {
    if (*maxcount > 127u) return;                         // gram_tile_kernel<Eucl> owns the matrix
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    const uint32_t wr = wave >> 1, wc = wave & 1;         // 2 x 2 waves of 64 x 64
    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM + wr * 64, j0 = (uint64_t)tj * TN + wc * 64;
    const double* T = rs;
    const double* inv = rs + A.npad;
    gram_i8_half<OUT>(A, c8, dpad, T, inv, i0, j0, false, lane & 31, lane >> 5);
    if (po_tile_mirrors(A, ti, tj))                       // G^T = B A^T: same stores, rows <-> columns
        gram_i8_half<OUT>(A, c8, dpad, T, inv, j0, i0, true, lane & 31, lane >> 5);
}

}  // namespace

size_t po_gram_i8_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return npad * po_round_up(dim, 32) + 2 * npad * sizeof(double) + 256;
}

// ws layout: int8 matrix | S, 1/n | maxcount
int po_launch_gram_i8_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                           uint64_t npad, void* ws, const uint32_t** flag_out) {
    const uint32_t dpad = (uint32_t)po_round_up(dim, 32);
    uint8_t* base = static_cast<uint8_t*>(ws);
    int8_t* c8 = reinterpret_cast<int8_t*>(base);
    double* rs = reinterpret_cast<double*>(base + npad * dpad);
    uint32_t* maxcount = reinterpret_cast<uint32_t*>(base + npad * dpad + 2 * npad * sizeof(double));
    PO_HIP(hipMemsetAsync(maxcount, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(prep_i8_kernel, dim3((uint32_t)((npad + 3) / 4)), dim3(256), 0, ctx->stream, d_counts,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, dim, dpad, c8, rs, npad, maxcount);
    PO_CHECK_LAUNCH("prep_i8_kernel");
    *flag_out = maxcount;
    return PO_OK;
}

int po_launch_gram_i8_tiles(po_ctx* ctx, const po_tile_args& a, const void* ws, uint64_t* tiles) {
    const uint32_t dpad = (uint32_t)po_round_up(a.dim, 32);
    const uint8_t* base = static_cast<const uint8_t*>(ws);
    const int8_t* c8 = reinterpret_cast<const int8_t*>(base);
    const double* rs = reinterpret_cast<const double*>(base + a.npad * dpad);
    const uint32_t* maxcount = reinterpret_cast<const uint32_t*>(base + a.npad * dpad + 2 * a.npad * sizeof(double));
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    if (a.out_f32)
        hipLaunchKernelGGL(gram_i8_tile_kernel<float>, dim3((uint32_t)nblocks), dim3(kThreads), 0, ctx->stream, a, c8, dpad, rs, maxcount);
    else
        hipLaunchKernelGGL(gram_i8_tile_kernel<double>, dim3((uint32_t)nblocks), dim3(kThreads), 0, ctx->stream, a, c8, dpad, rs, maxcount);
    PO_CHECK_LAUNCH("gram_i8_tile_kernel");
    return PO_OK;
}
