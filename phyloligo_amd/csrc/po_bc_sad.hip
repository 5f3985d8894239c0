// Stage 2, Bray-Curtis fast path for record pairs with EQUAL word totals: packed-byte SAD.
//
// Same quantity as valu_tile_kernel<BC> ('braycurtis' at
// /root/reference/phylopackage/bin/phyloligo.py:381 -> SciPy: sum|a-b| / sum|a+b|).  When two records have
// the same number of counted words n, |a_w - b_w| = |ca_w - cb_w| / n, so the numerator is an exact integer
// sum of absolute count differences.  With counts <= 255 four words fit one register and
// v_sad_u8 (sum of absolute differences of 4 byte pairs + accumulate) does four words per instruction:
// the per-pair cost drops from 2 float64 instructions per word to 1/4 integer instruction per word.
//
// Eligibility is per tile, decided on the device exactly as for the JSD table kernel: classify marks every
// block of 128 records with its common total (0 = mixed, empty, or a count above 255); a tile takes this
// path iff both classes are equal and non-zero, every other tile is left to valu_tile_kernel<BC>.
#include "po_tiles.h"

namespace {

constexpr int TM = 128, TN = 128;
constexpr int KC = 8;                                  // packed word groups (4 words each) per staging step
constexpr int kThreads = 256;
constexpr int kStageWords = KC * (TM + TN);            // uint32 per buffer

// P8t[g][npad] = bytes (c[4g], c[4g+1], c[4g+2], c[4g+3]) of record n, clamped to 255, zero padded
__global__ __launch_bounds__(256) void prep_pack8_kernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t dim,
                                                         uint32_t groups_pad, uint64_t npad, uint32_t* __restrict__ p8t,
                                                         uint32_t* __restrict__ maxcount) {
    __shared__ uint32_t tile[64][65];
    const uint64_t n0 = (uint64_t)blockIdx.x * 64;
    const uint32_t g0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bool vec = (dim & 3u) == 0u && (reinterpret_cast<uintptr_t>(counts) & 15u) == 0u;
    uint32_t mx = 0;
    for (uint32_t r = ty; r < 64; r += 4) {                        // r: record, tx: group
        const uint64_t row = n0 + r;
        uint32_t packed = 0;
        const uint32_t d = (g0 + tx) * 4;
        if (row < n) {
            uint32_t v4[4] = {0u, 0u, 0u, 0u};
            if (vec && d + 3 < dim) {                                   // one 16-byte load (rows of 4 k words from an aligned base)
                const uint4 q = *reinterpret_cast<const uint4*>(counts + row * dim + d);
                v4[0] = q.x; v4[1] = q.y; v4[2] = q.z; v4[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (d + e < dim) ? counts[row * dim + d + e] : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mx = max(mx, v4[e]);
                packed |= min(v4[e], 255u) << (8 * e);
            }
        }
        tile[r][tx] = packed;
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > *maxcount) atomicMax(maxcount, mx);
    __syncthreads();
    for (uint32_t r = ty; r < 64; r += 4)                          // r: group, tx: record
        if (g0 + r < groups_pad && n0 + tx < npad) p8t[(uint64_t)(g0 + r) * npad + n0 + tx] = tile[tx][r];
}

__global__ __launch_bounds__(128) void classify_bc_kernel(const unsigned long long* __restrict__ totals, uint64_t n,
                                                          const uint32_t* __restrict__ maxcount,
                                                          unsigned long long* __restrict__ cls) {
    const uint64_t r = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    const unsigned long long ref = totals[(uint64_t)blockIdx.x * 128];
    const bool ok = (r >= n) || (totals[r] == ref);
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) cls[blockIdx.x] = (all && ref > 0 && *maxcount <= 255u) ? ref : 0ull;
}


template <typename OUT>
__global__ __launch_bounds__(kThreads, 2) void bc_sad_tile_kernel(po_tile_args A, const uint32_t* __restrict__ p8t,
                                                                  uint32_t groups_pad,
                                                                  const unsigned long long* __restrict__ cls) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem);          // [2][A: KC x 128 | B: KC x 128]

    const uint32_t t = threadIdx.x;
    const uint32_t tx = t & 15, ty = t >> 4;
    const uint32_t lane = t & 63, wave = t >> 6;

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const unsigned long long ntot = cls[ti];
    if (ntot == 0 || cls[tj] != ntot) return;                      // valu_tile_kernel<BC> owns this tile
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;

    uint32_t acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0u;

    auto gstage = [&](uint32_t g0, uint32_t buf) {                  // wave w: groups 2w, 2w+1 of A and of B
        const uint32_t g = wave * 2 + (lane >> 5);
        const uint32_t* row = p8t + (uint64_t)(g0 + g) * A.npad + (lane & 31) * 4;
        uint32_t* dst = stage + buf * kStageWords + wave * 2 * TM;
        po_glds16(row + i0, dst);
        po_glds16(row + j0, dst + KC * TM);
    };
    gstage(0, 0);
    __syncthreads();

    auto double_sums = [&]() {                                      // folded operands (po_fold.hip)
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] <<= 1;
    };
    const uint32_t dbl_group = A.dbl_at == PO_NO_DOUBLING ? PO_NO_DOUBLING : A.dbl_at / 4;   // a multiple of KC (gran 32)
    uint32_t cur = 0;
    for (uint32_t g0 = 0; g0 < groups_pad; g0 += KC) {
        if (g0 + KC < groups_pad) gstage(g0 + KC, cur ^ 1);
        if (g0 == dbl_group) double_sums();
        const uint32_t* sA = stage + cur * kStageWords + ty * 8;
        const uint32_t* sB = stage + cur * kStageWords + KC * TM + tx * 2;
        // fragments of group k+1 are read from LDS while the 64 SADs of group k issue
        auto frag = [&](int k, uint32_t (&a)[8], uint32_t (&b)[8]) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(sA + k * TM);
            const uint4 a1 = *reinterpret_cast<const uint4*>(sA + k * TM + 4);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 bv = *reinterpret_cast<const uint2*>(sB + k * TN + 32 * q);
                b[2 * q] = bv.x; b[2 * q + 1] = bv.y;
            }
        };
        uint32_t fa[2][8], fb[2][8];
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (k + 1 < KC) frag(k + 1, fa[(k + 1) & 1], fb[(k + 1) & 1]);
#pragma unroll
            for (int ia = 0; ia < 8; ++ia)
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) acc[ia][ib] = __builtin_amdgcn_sad_u8(fa[k & 1][ia], fb[k & 1][ib], acc[ia][ib]);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (dbl_group != PO_NO_DOUBLING && dbl_group >= groups_pad) double_sums();

    // ---- epilogue: BC = (num / n) / (w_i + w_j) ------------------------------------------------------
    const double inv_n = 1.0 / (double)ntot;
    const double* st1 = A.rowstat + A.npad;        // sum f of each record
    double wi[8];
#pragma unroll
    for (int ia = 0; ia < 8; ++ia) wi[ia] = st1[i0 + ty * 8 + ia];
    double wj[8];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) wj[ib] = st1[min(j0 + 32 * (ib >> 1) + 2 * tx + (ib & 1), A.npad - 1)];
    // Every record of a profile matrix has sum f = 1 exactly (its counts add up to its total), so the denominator is 2 and the float64
    // division of every pair - a reciprocal and a dozen dependent instructions, a third of this kernel's vector work at 136 words - is a
    // scaling by 1/2, which gives the same bits.  Checked per wave on the terms themselves; anything else (empty records, edited
    // profiles) divides as before.
    bool unit = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) unit = unit && wi[q] == 1.0 && wj[q] == 1.0;
    const bool halve = __builtin_amdgcn_ballot_w64(!unit) == 0ull;
    const double half_inv_n = 0.5 * inv_n;
    double v[8][8];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) {
        const uint64_t j = min(j0 + 32 * (ib >> 1) + 2 * tx + (ib & 1), A.npad - 1);
#pragma unroll
        for (int ia = 0; ia < 8; ++ia) {
            const uint64_t i = i0 + ty * 8 + ia;
            const double x = halve ? (double)acc[ia][ib] * half_inv_n : ((double)acc[ia][ib] * inv_n) / (wi[ia] + wj[ib]);
            v[ia][ib] = (i == j) ? 0.0 : x;
        }
    }
    if constexpr (sizeof(OUT) == 4) po_store_block_f32<8, kThreads>(A, ti, tj, i0, j0, tx, ty, v, reinterpret_cast<float*>(smem));
    else po_store_block<OUT, 8, kThreads>(A, ti, tj, i0, j0, tx, ty, v, reinterpret_cast<double*>(smem));
}

}  // namespace

static inline uint32_t bc_groups_pad(uint32_t dim) { return (uint32_t)po_round_up((dim + 3) / 4, KC); }

size_t po_bc_sad_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return (size_t)bc_groups_pad(dim) * npad * sizeof(uint32_t) + npad / 128 * sizeof(unsigned long long) + 256;
}

void po_bc_sad_view(const void* ws, uint64_t npad, uint32_t dim, const uint32_t** p8t, uint32_t* groups_pad) {
    (void)npad;
    *p8t = static_cast<const uint32_t*>(ws);
    *groups_pad = bc_groups_pad(dim);
}

// ws layout: packed matrix | classes | maxcount
int po_launch_bc_sad_prep(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                          uint64_t npad, void* ws, const unsigned long long** cls_out) {
    const uint32_t gp = bc_groups_pad(dim);
    uint8_t* base = static_cast<uint8_t*>(ws);
    uint32_t* p8t = reinterpret_cast<uint32_t*>(base);
    unsigned long long* cls = reinterpret_cast<unsigned long long*>(base + (size_t)gp * npad * sizeof(uint32_t));
    uint32_t* maxcount = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(cls) + npad / 128 * sizeof(unsigned long long));
    PO_HIP(hipMemsetAsync(maxcount, 0, sizeof(uint32_t), ctx->stream));
    dim3 grid((uint32_t)(npad / 64), (gp + 63) / 64);
    hipLaunchKernelGGL(prep_pack8_kernel, grid, dim3(256), 0, ctx->stream, d_counts, n, dim, gp, npad, p8t, maxcount);
    PO_CHECK_LAUNCH("prep_pack8_kernel");
    hipLaunchKernelGGL(classify_bc_kernel, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, maxcount, cls);
    PO_CHECK_LAUNCH("classify_bc_kernel");
    *cls_out = cls;
    return PO_OK;
}

int po_launch_bc_sad_tiles(po_ctx* ctx, const po_tile_args& a, const void* ws, uint64_t* tiles) {
    const uint32_t gp = bc_groups_pad(a.dim);
    const uint8_t* base = static_cast<const uint8_t*>(ws);
    const uint32_t* p8t = reinterpret_cast<const uint32_t*>(base);
    const unsigned long long* cls = reinterpret_cast<const unsigned long long*>(base + (size_t)gp * a.npad * sizeof(uint32_t));
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = kMirrorLdsBytes > 2 * kStageWords * sizeof(uint32_t) ? (size_t)kMirrorLdsBytes : 2 * kStageWords * sizeof(uint32_t);
    if (a.out_f32)
        hipLaunchKernelGGL(bc_sad_tile_kernel<float>, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, p8t, gp, cls);
    else
        hipLaunchKernelGGL(bc_sad_tile_kernel<double>, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, p8t, gp, cls);
    PO_CHECK_LAUNCH("bc_sad_tile_kernel");
    return PO_OK;
}
