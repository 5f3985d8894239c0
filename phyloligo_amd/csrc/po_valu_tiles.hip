// Stage 2, elementwise-reduction metrics: JSD and Bray-Curtis tiles of the N x N matrix.
//
// Replaces one Python call per contig pair
//   phylodist.JSD / KL   (/root/reference/phylopackage/core/phylodist.py:18-24, :43-48)
//   'braycurtis'         (/root/reference/phylopackage/bin/phyloligo.py:381 -> SciPy)
// under sklearn's pairwise_distances (phyloligo.py:364-392).
//
// A 256-lane workgroup owns a 128 x 128 tile of pairs; each lane an 8 x 8 register block.
// Frequencies (float64, Ft[d][npad]) are staged 8 words at a time through a double-buffered
// LDS tile, so HBM/L2 sees each operand row once per tile and the kernel is bound by vector
// ALU issue, not by memory ("which roof binds").
//
// JSD uses the entropy decomposition  JSD(a,b) = 1/2 (E_a + E_b - S) + ln2/2 (w_a + w_b),
// E_x = sum x ln x and w_x = sum x (per row, po_prep.hip; w is 1 for a profile, 0 for an empty
// record), S = sum s ln s with s = a + b.  It has exactly the reference's masking semantics (terms with a zero
// numerator vanish, phylodist.py:22-24) and needs ONE logarithm per word and pair.  CDNA4 has
// no float64 log instruction, so ln s is a 128-interval table reduction done in the ALU:
//   s = 2^e m,  j = top 7 mantissa bits,  r = m*invc[j] - 1  (|r| <= 2^-8, one fma),
//   ln s = (e*ln2 + logc[j]) + (r - r^2/2 + r^3/3 - r^4/4)          (abs. error < 1e-12)
// The {invc, logc} pairs sit in LDS replicated 16x (one copy per bank quad) so that the
// per-lane lookups of a wave never conflict.
#include "po_internal.h"

#include <math.h>

namespace {

constexpr int TM = 128, TN = 128;     // tile of pairs per workgroup
constexpr int KC = 8;                 // words staged per step
constexpr int kThreads = 256;
constexpr int kTabEntries = 128;
constexpr int kTabBytes = kTabEntries * 256;            // 16 copies x 16 B per entry
constexpr int kStageDoubles = KC * (TM + TN);           // one buffer
constexpr double LN2 = 0.693147180559945309417232121458;

struct TileCoord { uint32_t ti, tj; };

// linear block id -> tile of the upper triangle (tj >= ti), row major
__device__ __forceinline__ TileCoord tri_decode(uint64_t b, uint32_t T) {
    const double tt = 2.0 * T + 1.0;
    uint32_t i = (uint32_t)((tt - sqrt(tt * tt - 8.0 * (double)b)) * 0.5);
    // rows before i hold i*T - i(i-1)/2 tiles; fix the float estimate
    auto before = [T](uint64_t r) { return r * T - r * (r - 1) / 2; };
    while (i > 0 && before(i) > b) --i;
    while (before((uint64_t)i + 1) <= b) ++i;
    return {i, (uint32_t)(i + (b - before(i)))};
}

template <typename T> __device__ __forceinline__ void store_out(void* out, uint64_t idx, double v);
template <> __device__ __forceinline__ void store_out<double>(void* out, uint64_t idx, double v) {
    static_cast<double*>(out)[idx] = v;
}
template <> __device__ __forceinline__ void store_out<float>(void* out, uint64_t idx, double v) {
    static_cast<float*>(out)[idx] = (float)v;
}

template <int METRIC, typename OUT>
__global__ __launch_bounds__(kThreads, 2) void valu_tile_kernel(po_tile_args A, const double2* __restrict__ logtab,
                                                                uint32_t tiles_n, uint32_t tile_row0) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* stage = reinterpret_cast<double*>(smem);                    // [2][KC][TM+TN]
    unsigned char* tab = smem + 2 * kStageDoubles * sizeof(double);     // JSD only

    const uint32_t t = threadIdx.x;
    const uint32_t tx = t & 15, ty = t >> 4;

    uint32_t ti, tj;
    if (A.symmetric) {
        const TileCoord c = tri_decode(blockIdx.x, tiles_n);
        ti = c.ti; tj = c.tj;
    } else {
        ti = tile_row0 + blockIdx.x / tiles_n;
        tj = blockIdx.x % tiles_n;
    }
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;

    if (METRIC == PO_JSD) {
        const uint4* src = reinterpret_cast<const uint4*>(logtab);       // already replicated, 32 KiB
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t v = t; v < kTabBytes / 16; v += kThreads) dst[v] = src[v];
    }

    double acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;

    // staging role: lane loads 4 consecutive records of word (k0 + sk) for A and for B
    const uint32_t sk = t >> 5, sc = (t & 31) * 4;
    const double* gA = A.ft + i0 + sc;
    const double* gB = A.ft + j0 + sc;
    double2 ra0, ra1, rb0, rb1;
    auto gload = [&](uint32_t k0) {
        const uint32_t k = k0 + sk;
        if (k < A.dim) {
            const double* pa = gA + (uint64_t)k * A.npad;
            const double* pb = gB + (uint64_t)k * A.npad;
            ra0 = *reinterpret_cast<const double2*>(pa);
            ra1 = *reinterpret_cast<const double2*>(pa + 2);
            rb0 = *reinterpret_cast<const double2*>(pb);
            rb1 = *reinterpret_cast<const double2*>(pb + 2);
        } else {
            ra0 = ra1 = rb0 = rb1 = make_double2(0.0, 0.0);
        }
    };
    auto sstore = [&](uint32_t buf) {
        double* s = stage + buf * kStageDoubles + sk * (TM + TN);
        *reinterpret_cast<double2*>(s + sc) = ra0;
        *reinterpret_cast<double2*>(s + sc + 2) = ra1;
        *reinterpret_cast<double2*>(s + TM + sc) = rb0;
        *reinterpret_cast<double2*>(s + TM + sc + 2) = rb1;
    };

    gload(0);
    sstore(0);
    __syncthreads();

    const uint32_t tcopy = tx * 16;                                      // this lane's table copy
    uint32_t cur = 0;
    for (uint32_t k0 = 0; k0 < A.dim; k0 += KC) {
        const bool more = k0 + KC < A.dim;
        if (more) gload(k0 + KC);
        const double* s = stage + cur * kStageDoubles;
#pragma unroll 2
        for (int k = 0; k < KC; ++k) {
            const double* sa = s + k * (TM + TN) + ty * 8;
            const double* sb = s + k * (TM + TN) + TM + tx * 2;
            double a[8], b[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 v = *reinterpret_cast<const double2*>(sa + 2 * q);
                a[2 * q] = v.x; a[2 * q + 1] = v.y;
                const double2 u = *reinterpret_cast<const double2*>(sb + 32 * q);
                b[2 * q] = u.x; b[2 * q + 1] = u.y;
            }
#pragma unroll
            for (int ia = 0; ia < 8; ++ia) {
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) {
                    if (METRIC == PO_JSD) {
                        const double sum = a[ia] + b[ib];
                        const uint32_t hi = (uint32_t)__double2hiint(sum);
                        const uint32_t lo = (uint32_t)__double2loint(sum);
                        const double2 te = *reinterpret_cast<const double2*>(tab + (((hi >> 5) & 0x7F00u) | tcopy));
                        const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);
                        const double X = __hiloint2double((int)(((hi >> 11) & 0x000FFE00u) | 0x40A00000u), 0);
                        const double r = fma(m, te.x, -1.0);
                        double q = fma(r, -0.25, 1.0 / 3.0);
                        q = fma(r, q, -0.5);
                        const double big = fma(X, LN2, te.y) + r;
                        const double ln_s = fma(r * r, q, big);
                        acc[ia][ib] = fma(sum, ln_s, acc[ia][ib]);
                    } else {  // PO_BC
                        acc[ia][ib] += fabs(a[ia] - b[ib]);
                    }
                }
            }
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const double* st0 = A.rowstat;             // sum f ln f
    const double* st1 = A.rowstat + A.npad;    // sum f
    double ei[8], wi[8];
#pragma unroll
    for (int ia = 0; ia < 8; ++ia) {
        const uint64_t i = i0 + ty * 8 + ia;
        ei[ia] = st0[i];
        wi[ia] = st1[i];
    }
    const bool mirror = A.symmetric && (ti != tj);
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) {
        const uint64_t j = j0 + 32 * (ib >> 1) + 2 * tx + (ib & 1);
        if (j >= A.n) continue;
        const double ej = st0[j], wj = st1[j];
#pragma unroll
        for (int ia = 0; ia < 8; ++ia) {
            const uint64_t i = i0 + ty * 8 + ia;
            if (i < A.row_begin || i >= A.row_end) continue;
            double v;
            if (METRIC == PO_JSD) {
                v = 0.5 * (ei[ia] + ej - acc[ia][ib]) + (0.5 * LN2) * (wi[ia] + wj);
                v = fmax(v, 0.0);
            } else {
                v = acc[ia][ib] / (wi[ia] + wj);                     // 0/0 -> NaN as SciPy gives
            }
            if (i == j) v = 0.0;                                     // metric(x,x) / squareform diagonal
            store_out<OUT>(A.out, (i - A.row_begin) * A.ld_out + j, v);
            if (mirror) store_out<OUT>(A.out, j * A.ld_out + i, v);
        }
    }
}

template <int METRIC>
int launch_metric(po_ctx* ctx, const po_tile_args& a, uint64_t* tiles) {
    const uint32_t T = (uint32_t)((a.n + TN - 1) / TN);
    uint64_t nblocks;
    uint32_t tile_row0 = 0;
    if (a.symmetric) {
        nblocks = (uint64_t)T * (T + 1) / 2;
    } else {
        tile_row0 = (uint32_t)(a.row_begin / TM);
        const uint32_t tile_row1 = (uint32_t)((a.row_end + TM - 1) / TM);
        nblocks = (uint64_t)(tile_row1 - tile_row0) * T;
    }
    if (tiles) *tiles = nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = 2 * kStageDoubles * sizeof(double) + (METRIC == PO_JSD ? kTabBytes : 0);
    const double2* tab = static_cast<const double2*>(ctx->ws_logtab.p);
    if (a.out_f32) {
        auto k = valu_tile_kernel<METRIC, float>;
        PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, tab, T, tile_row0);
    } else {
        auto k = valu_tile_kernel<METRIC, double>;
        PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, tab, T, tile_row0);
    }
    PO_CHECK_LAUNCH("valu_tile_kernel");
    return PO_OK;
}

}  // namespace

// {invc, logc - 3071 ln2} for the 128 mantissa intervals, each entry replicated 16x so that
// copy c of entry j sits at byte j*256 + c*16 (banks 4c..4c+3).
int po_logtab_init(po_ctx* ctx) {
    if (ctx->logtab_ready) return PO_OK;
    int rc = po_buf_reserve(ctx, &ctx->ws_logtab, kTabBytes);
    if (rc) return rc;
    static double host_tab[kTabEntries * 16 * 2];
    const long double ln2 = 0.693147180559945309417232121458176568L;
    for (int j = 0; j < kTabEntries; ++j) {
        const double c = 1.0 + (j + 0.5) / kTabEntries;
        const double invc = 1.0 / c;
        const long double logc = -logl((long double)invc) - 3071.0L * ln2;
        for (int r = 0; r < 16; ++r) {
            host_tab[(j * 16 + r) * 2 + 0] = invc;
            host_tab[(j * 16 + r) * 2 + 1] = (double)logc;
        }
    }
    PO_HIP(hipMemcpyAsync(ctx->ws_logtab.p, host_tab, kTabBytes, hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    ctx->logtab_ready = true;
    return PO_OK;
}

int po_launch_valu_tiles(po_ctx* ctx, int metric, const po_tile_args& a, uint64_t* tiles) {
    if (metric == PO_JSD) {
        int rc = po_logtab_init(ctx);
        if (rc) return rc;
        return launch_metric<PO_JSD>(ctx, a, tiles);
    }
    if (metric == PO_BC) return launch_metric<PO_BC>(ctx, a, tiles);
    po_set_error("po_launch_valu_tiles: metric %d is not an elementwise-reduction metric", metric);
    return PO_EINVAL;
}
