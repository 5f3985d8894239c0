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
#include "po_tiles.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int TM = 128, TN = 128;     // tile of pairs per workgroup
constexpr int KC = 8;                 // words staged per step
constexpr int kThreads = 256;
constexpr int kTabEntries = 128;
constexpr int kTabBytes = kTabEntries * 256;            // 16 copies x 16 B per entry
constexpr int kStageDoubles = KC * (TM + TN);           // one buffer
constexpr double LN2 = 0.693147180559945309417232121458;




struct JsdConsts {
    uint32_t tcopy;   // LDS byte address of this lane's table copy
    uint32_t k3ff;    // 0x3ff00000 in a VGPR (VOP3 takes no literal on gfx9)
    double c4;        // -1/4 in a VGPR pair
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef const __attribute__((address_space(1))) unsigned char glb_byte;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)((const lds_byte*)p); }
__device__ __forceinline__ double2 lds_read_d2(uint32_t addr) {
    return *((const __attribute__((address_space(3))) double2*)(uintptr_t)addr);
}
// 64 lanes x 16 B straight from global memory into 1 KiB of LDS at `lds_base` (wave uniform)
__device__ __forceinline__ void glds16(const void* gptr, void* lds_base) {
    __builtin_amdgcn_global_load_lds((glb_byte*)gptr, (lds_byte*)lds_base, 16, 0, 0);
}
#else
__device__ __forceinline__ uint32_t lds_addr(const void*) { return 0; }
__device__ __forceinline__ double2 lds_read_d2(uint32_t) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ void glds16(const void*, void*) {}
#endif

constexpr int GS = 2;              // pairs per pipeline group
constexpr int NG = 64 / GS;        // groups per word

// fragments of one staged word: 8 records of the row block (broadcast), 8 of the column block.
// HEAD = what the first 8 groups of the word need (a[0..7] come in pairs: a[0..1], b[0..1]);
// REST = everything else.
template <bool HEAD, bool REST>
__device__ __forceinline__ void load_frag(const double* s, int k, uint32_t tx, uint32_t ty, double (&a)[8], double (&b)[8]) {
    const double* sa = s + k * (TM + TN) + ty * 8;
    const double* sb = s + k * (TM + TN) + TM + tx * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if ((q == 0) ? HEAD : REST) {
            const double2 v = *reinterpret_cast<const double2*>(sa + 2 * q);
            a[2 * q] = v.x; a[2 * q + 1] = v.y;
            const double2 u = *reinterpret_cast<const double2*>(sb + 32 * q);
            b[2 * q] = u.x; b[2 * q + 1] = u.y;
        }
    }
}

// group G of a word: pairs (ia = G%8, ib = GS*(G/8) .. +GS-1).  Sum, table address, table read.
template <int G>
__device__ __forceinline__ void jsd_issue(const JsdConsts& C, const double (&a)[8], const double (&b)[8],
                                          double (&psum)[GS], double2 (&pte)[GS]) {
    constexpr int ia = G & 7, ib0 = (G >> 3) * GS;
#pragma unroll
    for (int e = 0; e < GS; ++e) {
        const double sum = a[ia] + b[ib0 + e];
        uint32_t toff;
        asm("v_bfe_u32 %0, %1, 13, 7\n\tv_lshl_or_b32 %0, %0, 8, %2" : "=&v"(toff) : "v"(__double2hiint(sum)), "v"(C.tcopy));
        psum[e] = sum;
        pte[e] = lds_read_d2(toff);
    }
}

template <int VAR, int G>
__device__ __forceinline__ void jsd_eval(const JsdConsts& C, const double (&psum)[GS], const double2 (&pte)[GS],
                                         double (&acc)[8][8]) {
    constexpr int ia = G & 7, ib0 = (G >> 3) * GS;
    const double c3 = 1.0 / 3.0;
#pragma unroll
    for (int e = 0; e < GS; ++e) {
        const double sum = psum[e];
        const double2 te = pte[e];
        double m, ef;
        if (VAR == 0) {          // integer field surgery on the bit pattern
            const uint32_t hi = (uint32_t)__double2hiint(sum);
            uint32_t mhi;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(mhi) : "v"(hi), "s"(0x000FFFFFu), "v"(C.k3ff));
            m = __hiloint2double((int)mhi, __double2loint(sum));
            ef = (double)__builtin_amdgcn_ubfe(hi, 20, 11);
        } else {                 // hardware frexp
            m = __builtin_amdgcn_frexp_mant(sum);
            ef = (double)__builtin_amdgcn_frexp_exp(sum);
        }
        double r, q;
        asm("v_fma_f64 %0, %1, %2, -1.0" : "=v"(r) : "v"(m), "v"(te.x));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(r), "v"(C.c4), "s"(c3));
        q = fma(r, q, -0.5);
        q = fma(r, q, 1.0);
        const double big = fma(ef, LN2, te.y);
        const double ln_s = fma(r, q, big);
        acc[ia][ib0 + e] = fma(sum, ln_s, acc[ia][ib0 + e]);
    }
}

__device__ __forceinline__ void jsd_issue_ab(const JsdConsts& C, double a, const double* b, double (&psum)[GS], double2 (&pte)[GS]) {
#pragma unroll
    for (int e = 0; e < GS; ++e) {
        const double sum = a + b[e];
        uint32_t toff;
        asm("v_bfe_u32 %0, %1, 13, 7\n\tv_lshl_or_b32 %0, %0, 8, %2" : "=&v"(toff) : "v"(__double2hiint(sum)), "v"(C.tcopy));
        psum[e] = sum;
        pte[e] = lds_read_d2(toff);
    }
}

template <int VAR>
__device__ __forceinline__ void jsd_eval_ab(const JsdConsts& C, const double (&psum)[GS], const double2 (&pte)[GS], double* acc) {
    const double c3 = 1.0 / 3.0;
#pragma unroll
    for (int e = 0; e < GS; ++e) {
        const double sum = psum[e];
        const double2 te = pte[e];
        double m, ef;
        if (VAR == 0) {
            const uint32_t hi = (uint32_t)__double2hiint(sum);
            uint32_t mhi;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(mhi) : "v"(hi), "s"(0x000FFFFFu), "v"(C.k3ff));
            m = __hiloint2double((int)mhi, __double2loint(sum));
            ef = (double)__builtin_amdgcn_ubfe(hi, 20, 11);
        } else {
            m = __builtin_amdgcn_frexp_mant(sum);
            ef = (double)__builtin_amdgcn_frexp_exp(sum);
        }
        double r, q;
        asm("v_fma_f64 %0, %1, %2, -1.0" : "=v"(r) : "v"(m), "v"(te.x));
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(r), "v"(C.c4), "s"(c3));
        q = fma(r, q, -0.5);
        q = fma(r, q, 1.0);
        const double big = fma(ef, LN2, te.y);
        const double ln_s = fma(r, q, big);
        acc[e] = fma(sum, ln_s, acc[e]);
    }
}

// Groups G..NG-1 of one word: while group G is evaluated from buffer G%2, group G+1 (or group 0 of
// the next word) is in flight in the other buffer.
template <int VAR, bool NEXT, int G>
__device__ __forceinline__ void jsd_groups(const JsdConsts& C, const double* s, int k, uint32_t tx, uint32_t ty,
                                           const double (&a)[8], const double (&b)[8], double (&an)[8], double (&bn)[8],
                                           double (&p0)[GS], double2 (&t0)[GS], double (&p1)[GS], double2 (&t1)[GS],
                                           double (&acc)[8][8]) {
    if constexpr (G == NG - 8) {
        if (NEXT) load_frag<true, false>(s, k + 1, tx, ty, an, bn);
    }
    if constexpr (G + 1 < NG) {
        if constexpr ((G & 1) == 0) jsd_issue<G + 1>(C, a, b, p1, t1); else jsd_issue<G + 1>(C, a, b, p0, t0);
    } else if (NEXT) {
        if constexpr ((G & 1) == 0) jsd_issue<0>(C, an, bn, p1, t1); else jsd_issue<0>(C, an, bn, p0, t0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((G & 1) == 0) jsd_eval<VAR, G>(C, p0, t0, acc); else jsd_eval<VAR, G>(C, p1, t1, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < NG) jsd_groups<VAR, NEXT, G + 1>(C, s, k, tx, ty, a, b, an, bn, p0, t0, p1, t1, acc);
}

// One word (64 pairs) at staged position k.  On entry the HEAD fragments of the word are loaded and
// its group 0 is in flight in (p0,t0); on exit, when NEXT, the same holds for word k+1 (an, bn).
// NG is even, so the buffer roles are the same for every word.
template <int VAR, bool NEXT>
__device__ __forceinline__ void jsd_word(const JsdConsts& C, const double* s, int k, uint32_t tx, uint32_t ty,
                                         double (&a)[8], double (&b)[8], double (&an)[8], double (&bn)[8],
                                         double (&p0)[GS], double2 (&t0)[GS], double (&p1)[GS], double2 (&t1)[GS],
                                         double (&acc)[8][8]) {
    load_frag<false, true>(s, k, tx, ty, a, b);
    jsd_groups<VAR, NEXT, 0>(C, s, k, tx, ty, a, b, an, bn, p0, t0, p1, t1, acc);
}

template <int METRIC, typename OUT, int VAR>
__global__ __launch_bounds__(kThreads, 2) void valu_tile_kernel(po_tile_args A, const double2* __restrict__ logtab,
                                                                const unsigned long long* __restrict__ cls) {
    extern __shared__ __align__(16) unsigned char smem[];
    // the log table comes first so that a lane's lookup address is just (interval << 8 | copy)
    unsigned char* tab = smem;                                          // JSD only, kTabBytes
    double* stage = reinterpret_cast<double*>(smem + (METRIC == PO_JSD ? kTabBytes : 0));   // [2][KC][TM+TN]

    const uint32_t t = threadIdx.x;
    const uint32_t tx = t & 15, ty = t >> 4;

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    if (cls != nullptr) {      // tiles of equal-total record blocks belong to po_jsd_lut.hip / po_bc_sad.hip
        const unsigned long long c = cls[ti];
        if (c != 0 && cls[tj] == c) return;
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

    // staging: wave w copies words 2w, 2w+1 of the step -- 64 lanes x 16 B per instruction, straight
    // from HBM/L2 into the lane-linear LDS rows (no VGPR round trip)
    const uint32_t lane = t & 63, wave = t >> 6;
    auto gstage = [&](uint32_t k0, uint32_t buf) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t k = wave * 2 + r;
            const double* row = A.ft + (uint64_t)(k0 + k) * A.npad + lane * 2;
            double* dst = stage + buf * kStageDoubles + k * (TM + TN);
            glds16(row + i0, dst);
            glds16(row + j0, dst + TM);
        }
    };
    gstage(0, 0);
    __syncthreads();

    // this lane's copy of the table as a raw LDS byte address: lookup = (interval << 8) | tcopy
    JsdConsts C;
    C.tcopy = lds_addr(tab) + tx * 16;
    {
        uint32_t c4lo, c4hi;
        asm volatile("v_mov_b32 %0, 0x3ff00000" : "=v"(C.k3ff));
        asm volatile("v_mov_b32 %0, 0" : "=v"(c4lo));
        asm volatile("v_mov_b32 %0, 0xbfd00000" : "=v"(c4hi));
        C.c4 = __hiloint2double((int)c4hi, (int)c4lo);                  // -1/4 held in a VGPR pair
    }

    uint32_t cur = 0;
    for (uint32_t k0 = 0; k0 < A.dim; k0 += KC) {
        const bool more = k0 + KC < A.dim;
        if (more) gstage(k0 + KC, cur ^ 1);
        const double* s = stage + cur * kStageDoubles;
        if (METRIC == PO_JSD) {
#pragma unroll 2
            for (int k = 0; k < KC; ++k) {
                double a[8], b[8];
                load_frag<true, true>(s, k, tx, ty, a, b);
                // one-deep software pipeline over the 32 groups of the word: the table reads of
                // group g+1 are issued before group g is evaluated
                double psum[2][GS];
                double2 pte[2][GS];
                jsd_issue_ab(C, a[0], &b[0], psum[0], pte[0]);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int gn = g + 1;
                    if (gn < NG) jsd_issue_ab(C, a[gn >> 2], &b[(gn & 3) * GS], psum[gn & 1], pte[gn & 1]);
                    jsd_eval_ab<VAR>(C, psum[g & 1], pte[g & 1], &acc[g >> 2][(g & 3) * GS]);
                }
            }
        } else {  // PO_BC
#pragma unroll 2
            for (int k = 0; k < KC; ++k) {
                double a[8], b[8];
                load_frag<true, true>(s, k, tx, ty, a, b);
#pragma unroll
                for (int ia = 0; ia < 8; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 8; ++ib) acc[ia][ib] += fabs(a[ia] - b[ib]);
            }
        }
        __syncthreads();      // also drains this wave's in-flight LDS-DMA (vmcnt) before the buffers swap
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
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) {
        const uint64_t j = min(j0 + 32 * (ib >> 1) + 2 * tx + (ib & 1), A.npad - 1);
        const double ej = st0[j], wj = st1[j];
#pragma unroll
        for (int ia = 0; ia < 8; ++ia) {
            const uint64_t i = i0 + ty * 8 + ia;
            double v;
            if (METRIC == PO_JSD) {
                v = 0.5 * (ei[ia] + ej - acc[ia][ib]) + (0.5 * LN2) * (wi[ia] + wj);
                v = fmax(v, 0.0);
            } else {
                v = acc[ia][ib] / (wi[ia] + wj);                     // 0/0 -> NaN as SciPy gives
            }
            if (i == j) v = 0.0;                                     // metric(x,x) / squareform diagonal
            acc[ia][ib] = v;
        }
    }
    po_store_block<OUT, 8, kThreads>(A, ti, tj, i0, j0, tx, ty, acc, reinterpret_cast<double*>(smem));
}

template <int METRIC, int VAR>
int launch_metric(po_ctx* ctx, const po_tile_args& a, const unsigned long long* cls, uint64_t* tiles) {
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = max((size_t)kMirrorLdsBytes, 2 * kStageDoubles * sizeof(double) + (METRIC == PO_JSD ? kTabBytes : 0));
    const double2* tab = reinterpret_cast<const double2*>(static_cast<const unsigned char*>(ctx->ws_logtab.p) + VAR * kTabBytes);
    if (a.out_f32) {
        auto k = valu_tile_kernel<METRIC, float, VAR>;
        PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, tab, cls);
    } else {
        auto k = valu_tile_kernel<METRIC, double, VAR>;
        PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, tab, cls);
    }
    PO_CHECK_LAUNCH("valu_tile_kernel");
    return PO_OK;
}

}  // namespace

// Log tables for the 128 mantissa intervals, each entry replicated 16x so that copy c of entry j
// sits at byte j*256 + c*16 (banks 4c..4c+3).  Two layouts, one per reduction variant:
//   0: s = 2^(eb-1023) m, m in [1,2):  {invc,   -ln(invc) - 1023 ln2},  ef = (double)eb
//   1: s = 2^e m,         m in [.5,1): {2 invc, -ln(invc) - ln2},       ef = (double)e
int po_logtab_init(po_ctx* ctx) {
    if (ctx->logtab_ready) return PO_OK;
    int rc = po_buf_reserve(ctx, &ctx->ws_logtab, 2 * kTabBytes);
    if (rc) return rc;
    static double host_tab[2][kTabEntries * 16 * 2];
    const long double ln2 = 0.693147180559945309417232121458176568L;
    for (int j = 0; j < kTabEntries; ++j) {
        const double c = 1.0 + (j + 0.5) / kTabEntries;
        const double invc = 1.0 / c;
        const long double logc = -logl((long double)invc);
        for (int r = 0; r < 16; ++r) {
            host_tab[0][(j * 16 + r) * 2 + 0] = invc;
            host_tab[0][(j * 16 + r) * 2 + 1] = (double)(logc - 1023.0L * ln2);
            host_tab[1][(j * 16 + r) * 2 + 0] = 2.0 * invc;
            host_tab[1][(j * 16 + r) * 2 + 1] = (double)(logc - ln2);
        }
    }
    PO_HIP(hipMemcpyAsync(ctx->ws_logtab.p, host_tab, 2 * kTabBytes, hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    ctx->logtab_ready = true;
    return PO_OK;
}

int po_launch_valu_tiles(po_ctx* ctx, int metric, const po_tile_args& a, const unsigned long long* cls, uint64_t* tiles) {
    if (metric == PO_JSD) {
        int rc = po_logtab_init(ctx);
        if (rc) return rc;
        static const int variant = getenv("PO_JSD_VARIANT") ? atoi(getenv("PO_JSD_VARIANT")) : 0;
        if (variant == 1) return launch_metric<PO_JSD, 1>(ctx, a, cls, tiles);
        return launch_metric<PO_JSD, 0>(ctx, a, cls, tiles);
    }
    if (metric == PO_BC) return launch_metric<PO_BC, 0>(ctx, a, cls, tiles);
    po_set_error("po_launch_valu_tiles: metric %d is not an elementwise-reduction metric", metric);
    return PO_EINVAL;
}
