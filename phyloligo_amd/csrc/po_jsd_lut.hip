// Stage 2, JSD fast path for record pairs with EQUAL word totals: integer-sum table lookup.
//
// Same mathematics as valu_tile_kernel<JSD> (phylodist.JSD / KL,
// /root/reference/phylopackage/core/phylodist.py:18-24, :43-48).  When two records have the same number
// of counted words n (fixed-length contigs, sliding windows, simulated reads), the mixture
// h = (a+b)/2 has h_w = (ca_w + cb_w) / 2n, so
//     S = sum_w (a_w+b_w) ln(a_w+b_w) = (1/n) sum_w T[ca_w + cb_w] - 2 ln n,   T[x] = x ln x,
// and the per-word, per-pair work collapses from a float64 logarithm to ONE integer add, ONE LDS read
// and ONE float64 add.  T[x] for x = 0..255 sits in LDS replicated 32x (entry x, copy c at byte
// x*256 + c*8) so that the per-lane lookups of a wave never conflict; counts are staged pre-shifted by 8
// bits, so a lookup address is a single v_add3_u32.
//
// Eligibility is decided per tile on the device: classify_kernel marks each block of 128 records with its
// common total (0 = mixed / empty / a count above 127); a tile (I,J) takes this path iff both classes are
// equal and non-zero, every other tile is left to valu_tile_kernel<JSD>, which skips the marked ones.
#include "po_tiles.h"

#include <stdlib.h>

namespace {

constexpr int TM = 128, TN = 128;
constexpr int KC = 8;
constexpr int kLutEntries = 256;                       // sums 0..255  -> counts up to 127 (fixed-length contigs up to ~10 kb at k=4)
constexpr int kLutBytes = kLutEntries * 256;           // 32 copies x 8 B per entry
constexpr int kStageWords = KC * (TM + TN);            // uint32 per buffer
constexpr double LN2 = 0.693147180559945309417232121458;

// Ct[d][npad] = counts[n][d] << 8 (uint32, transposed, zero padded to D8 x npad)
__global__ __launch_bounds__(256) void prep_counts_kernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t dim,
                                                          uint64_t npad, uint32_t* __restrict__ ct,
                                                          uint32_t* __restrict__ maxcount) {
    __shared__ uint32_t tile[64][65];
    __shared__ uint32_t blkmax;
    const uint64_t n0 = (uint64_t)blockIdx.x * 64;
    const uint32_t d0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x == 0) blkmax = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint64_t row = n0 + r;
        const uint32_t v = (row < n && d0 + tx < dim) ? counts[row * dim + d0 + tx] : 0u;
        mx = max(mx, v);
        tile[r][tx] = v << 8;
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&blkmax, mx);
    __syncthreads();
    if (threadIdx.x == 0 && blkmax > *maxcount) atomicMax(maxcount, blkmax);   // racy pre-check skips redundant atomics
    for (uint32_t r = ty; r < 64; r += 4)
        if (d0 + r < ((dim + 7u) & ~7u) && n0 + tx < npad) ct[(uint64_t)(d0 + r) * npad + n0 + tx] = tile[tx][r];
}

// cls[b] = common word total of records [128b, 128b+128) (padding ignored), 0 if they differ, if one is
// empty, or if any count in the matrix exceeds what the table covers.
// The table identity needs sum_w count = total for every record (what count2freq guarantees); a caller-supplied
// total that disagrees with its counts shows up as wsum = sum_w count/total != 1 and sends the block to the general kernel.
__global__ __launch_bounds__(128) void classify_kernel(const unsigned long long* __restrict__ totals, uint64_t n,
                                                       const uint32_t* __restrict__ maxcount,
                                                       const double* __restrict__ wsum,
                                                       unsigned long long* __restrict__ cls) {
    const uint64_t r = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * 128;
    const unsigned long long ref = totals[first];                     // first < n by construction of the grid
    const bool ok = (r >= n) || (totals[r] == ref && fabs(wsum[r] - 1.0) < 1e-9);
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) cls[blockIdx.x] = (all && ref > 0 && 2u * *maxcount < (uint32_t)kLutEntries) ? ref : 0ull;
}

// the table as the tile kernels want it in LDS: entry x replicated 32 times (copy c at x*32 + c)
__global__ void lut_table_kernel(double* __restrict__ lut) {
    const uint32_t x = blockIdx.x;
    lut[x * 32 + threadIdx.x] = x ? (double)x * log((double)x) : 0.0;
}



// RPT rows per lane: 8 -> 256 lanes per 128 x 128 tile, 4 -> 512 lanes (more waves per SIMD to hide the
// LDS round trips; the table is shared by twice as many waves).
template <typename OUT, int RPT>
__global__ __launch_bounds__(2048 / RPT, RPT == 8 ? 2 : 4) void jsd_lut_tile_kernel(po_tile_args A, const uint32_t* __restrict__ ct,
                                                                                  const double* __restrict__ lut,
                                                                                  const unsigned long long* __restrict__ cls) {
    constexpr int NT = 2048 / RPT;                                               // lanes per workgroup
    extern __shared__ __align__(16) unsigned char smem[];
    double* tab = reinterpret_cast<double*>(smem);                               // [128][32]
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem + kLutBytes);             // [2][A: KC x 128 | B: KC x 128]

    const uint32_t t = threadIdx.x;
    const uint32_t tx = t & 15, ty = t >> 4;
    const uint32_t lane = t & 63, wave = t >> 6;

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const unsigned long long ntot = cls[ti];
    if (ntot == 0 || cls[tj] != ntot) return;                                    // valu_tile_kernel<JSD> owns this tile
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;

    {
        const uint4* src = reinterpret_cast<const uint4*>(lut);                  // already replicated: 16-byte copies
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t v = t; v < kLutBytes / 16; v += NT) dst[v] = src[v];
    }

    double acc[RPT][8];
#pragma unroll
    for (int a = 0; a < RPT; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;

    // staging: one LDS-DMA instruction moves 64 lanes x 16 B = two 512-byte word rows.  With 4 waves
    // each wave takes words 2w, 2w+1 of the A block and of the B block; with 8 waves, waves 0-3 take
    // the A block and waves 4-7 the B block.
    auto gstage = [&](uint32_t k0, uint32_t buf) {
        const uint32_t w4 = wave & 3;
        const uint32_t k = w4 * 2 + (lane >> 5);
        const uint32_t* row = ct + (uint64_t)(k0 + k) * A.npad + (lane & 31) * 4;
        uint32_t* dst = stage + buf * kStageWords + w4 * 2 * TM;
        if (RPT == 8) {
            po_glds16(row + i0, dst);
            po_glds16(row + j0, dst + KC * TM);
        } else if (wave < 4) {
            po_glds16(row + i0, dst);
        } else {
            po_glds16(row + j0, dst + KC * TM);
        }
    };
    gstage(0, 0);
    __syncthreads();

    const uint32_t tcopy = po_lds_addr(tab) + (lane & 31) * 8;
    auto double_sums = [&]() {                                                   // folded operands (po_fold.hip)
#pragma unroll
        for (int a = 0; a < RPT; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] *= 2.0;
    };
    uint32_t cur = 0;
    for (uint32_t k0 = 0; k0 < A.dim; k0 += KC) {
        if (k0 + KC < A.dim) gstage(k0 + KC, cur ^ 1);
        if (k0 == A.dbl_at) double_sums();
        const uint32_t* sA = stage + cur * kStageWords + ty * RPT;
        const uint32_t* sB = stage + cur * kStageWords + KC * TM + tx * 2;
#pragma unroll 2
        for (int k = 0; k < KC; ++k) {
            uint32_t a[RPT];                               // table base (this lane's copy) folded into the row side: RPT adds, not 8
#pragma unroll
            for (int q = 0; q < RPT / 4; ++q) {
                const uint4 av = *reinterpret_cast<const uint4*>(sA + k * TM + 4 * q);
                a[4 * q] = av.x + tcopy; a[4 * q + 1] = av.y + tcopy; a[4 * q + 2] = av.z + tcopy; a[4 * q + 3] = av.w + tcopy;
            }
            uint32_t b[8];                                 // columns 32*q + 2*tx + {0,1}
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 bv = *reinterpret_cast<const uint2*>(sB + k * TN + 32 * q);
                b[2 * q] = bv.x; b[2 * q + 1] = bv.y;
            }
            // the 8 lookups of register-block row ia+1 are in flight while row ia is accumulated
            double tv[2][8];
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) tv[0][ib] = po_lds_read_f64(a[0] + b[ib]);
#pragma unroll
            for (int ia = 0; ia < RPT; ++ia) {
                if (ia + 1 < RPT) {
#pragma unroll
                    for (int ib = 0; ib < 8; ++ib) tv[(ia + 1) & 1][ib] = po_lds_read_f64(a[ia + 1] + b[ib]);
                }
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) acc[ia][ib] += tv[ia & 1][ib];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (A.dbl_at != PO_NO_DOUBLING && A.dbl_at >= A.dim) double_sums();

    // ---- epilogue: JSD = 1/2 (E_i + E_j - S) + ln 2,  S = Tsum/n - 2 ln n ----------------------------
    const double inv_n = 1.0 / (double)ntot;
    const double two_ln_n = 2.0 * log((double)ntot);
    const double* st0 = A.rowstat;
    double ei[RPT];
#pragma unroll
    for (int ia = 0; ia < RPT; ++ia) ei[ia] = st0[i0 + ty * RPT + ia];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) {
        const uint64_t j = min(j0 + 32 * (ib >> 1) + 2 * tx + (ib & 1), A.npad - 1);
        const double ej = st0[j];
#pragma unroll
        for (int ia = 0; ia < RPT; ++ia) {
            const uint64_t i = i0 + ty * RPT + ia;
            const double S = fma(acc[ia][ib], inv_n, -two_ln_n);
            double v = fmax(0.5 * (ei[ia] + ej - S) + LN2, 0.0);
            if (i == j) v = 0.0;
            acc[ia][ib] = v;
        }
    }
    po_store_block<OUT, RPT, NT>(A, ti, tj, i0, j0, tx, ty, acc, reinterpret_cast<double*>(smem));
}

}  // namespace

size_t po_jsd_lut_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return po_round_up(dim, 8) * npad * sizeof(uint32_t)      // Ct
           + po_round_up(npad / 128 * sizeof(unsigned long long), 16)   // cls
           + kLutBytes + 256;                                   // replicated T + maxcount
}

// Builds Ct, the table and the tile classes in ws (layout as sized above); returns the class array.
int po_launch_jsd_lut_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                           uint64_t npad, const double* d_wsum, void* ws, const unsigned long long** cls_out) {
    uint8_t* base = static_cast<uint8_t*>(ws);
    uint32_t* ct = reinterpret_cast<uint32_t*>(base);
    base += po_round_up(dim, 8) * npad * sizeof(uint32_t);
    unsigned long long* cls = reinterpret_cast<unsigned long long*>(base);
    base += po_round_up(npad / 128 * sizeof(unsigned long long), 16);
    double* lut = reinterpret_cast<double*>(base);
    uint32_t* maxcount = reinterpret_cast<uint32_t*>(base + kLutBytes);
    PO_HIP(hipMemsetAsync(maxcount, 0, sizeof(uint32_t), ctx->stream));
    dim3 grid((uint32_t)(npad / 64), (dim + 63) / 64);
    hipLaunchKernelGGL(prep_counts_kernel, grid, dim3(256), 0, ctx->stream, d_counts, n, dim, npad, ct, maxcount);
    PO_CHECK_LAUNCH("prep_counts_kernel");
    hipLaunchKernelGGL(lut_table_kernel, dim3(kLutEntries), dim3(32), 0, ctx->stream, lut);
    PO_CHECK_LAUNCH("lut_table_kernel");
    hipLaunchKernelGGL(classify_kernel, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, maxcount, d_wsum, cls);
    PO_CHECK_LAUNCH("classify_kernel");
    *cls_out = cls;
    return PO_OK;
}

int po_launch_jsd_lut_tiles(po_ctx* ctx, const po_tile_args& a, uint64_t n, const void* ws, uint64_t* tiles) {
    const uint8_t* base = static_cast<const uint8_t*>(ws);
    const uint32_t* ct = reinterpret_cast<const uint32_t*>(base);
    base += po_round_up(a.dim, 8) * a.npad * sizeof(uint32_t);
    const unsigned long long* cls = reinterpret_cast<const unsigned long long*>(base);
    base += po_round_up(a.npad / 128 * sizeof(unsigned long long), 16);
    const double* lut = reinterpret_cast<const double*>(base);
    (void)n;
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = kLutBytes + 2 * kStageWords * sizeof(uint32_t);
    // 4 rows per lane (512 lanes per tile); the 8-row / 256-lane variant measured slower (round 1)
    if (a.out_f32) {
        PO_SHMEM(ctx, (jsd_lut_tile_kernel<float, 4>), shmem);
        hipLaunchKernelGGL((jsd_lut_tile_kernel<float, 4>), dim3((uint32_t)nblocks), dim3(512), shmem, ctx->stream, a, ct, lut, cls);
    } else {
        PO_SHMEM(ctx, (jsd_lut_tile_kernel<double, 4>), shmem);
        hipLaunchKernelGGL((jsd_lut_tile_kernel<double, 4>), dim3((uint32_t)nblocks), dim3(512), shmem, ctx->stream, a, ct, lut, cls);
    }
    PO_CHECK_LAUNCH("jsd_lut_tile_kernel");
    return PO_OK;
}
