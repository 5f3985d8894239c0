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
#include <type_traits>

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
        if (SIGNED) {                                     // Spearman: N = |r2|^2 and 1 / sqrt(N) (inf for a constant record: NaN later, as SciPy gives)
            rs[r] = (double)sq_s[rr];
            rs[npad + r] = 1.0 / sqrt((double)sq_s[rr]);
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

// ---- which tiles a launch owns (round 5) --------------------------------------------------------------------------------
// idx == nullptr: all `count` tiles of the block, workgroup (or virtual workgroup) w -> the XCD-banded order of po_tiles.h.
// idx != nullptr: the `count` logical positions listed there - the tiles of ONE class (how many digit planes their two record
// blocks need), built by classify_tiles_kernel in ascending order inside chunks of 256 positions; po_xcd_swizzle over the list
// gives every XCD one contiguous piece of it, so the tiles an XCD works on together still share their operands.
struct tile_list {
    const uint32_t* idx;
    uint64_t count;
};
__device__ __forceinline__ void list_tile(const po_tile_args& A, const tile_list& tl, uint64_t w, uint32_t& ti, uint32_t& tj) {
    uint64_t L = po_xcd_swizzle(w, tl.count);
    if (tl.idx != nullptr) L = tl.idx[L];
    po_tile_coords_logical(A, TM, L, ti, tj);
}

__device__ __forceinline__ uint32_t plane_class(uint32_t m) { return m <= 127u ? 0u : (m <= 16383u ? 1u : 2u); }

// lists[c][..] = logical positions of the block's tiles whose two record blocks need c + 1 digit planes; counts[c] their number
// (zeroed by the caller).  A workgroup takes 256 consecutive positions and reserves its piece of every list with one atomic add.
__global__ __launch_bounds__(256) void classify_tiles_kernel(po_tile_args A, const uint32_t* __restrict__ maxabs, uint64_t nb,
                                                             uint32_t* __restrict__ lists, uint32_t* __restrict__ counts) {
    __shared__ uint32_t wave_n[3][4], base[3];
    const uint64_t L = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cls = 3;
    if (L < nb) {
        uint32_t ti, tj;
        po_tile_coords_logical(A, TM, L, ti, tj);
        cls = max(plane_class(maxabs[1 + ti]), plane_class(maxabs[1 + tj]));
    }
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == c);
        if (cls == c) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_n[c][wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t c = threadIdx.x, total = wave_n[c][0] + wave_n[c][1] + wave_n[c][2] + wave_n[c][3];
        base[c] = total ? atomicAdd(&counts[c], total) : 0u;
    }
    __syncthreads();
    if (cls < 3) {
        uint32_t at = base[cls] + rank;
        for (uint32_t w = 0; w < wave; ++w) at += wave_n[cls][w];
        lists[(size_t)cls * nb + at] = (uint32_t)L;
    }
}

// The distance of a row record and a column record from their exact integer Gram entry G (shared by both tile kernels, so
// that every path gives the same bits).  Per-record terms - Eucl: t0 = S/n^2, t1 = 1/n, t2 = S (exact);  SC: t0 = N, t1 = 1/sqrt(N).
//   Eucl: d^2 = (t0_r + t0_c) + G * ((-2 t1_r) * t1_c): the doubling is exact, so the product of the two per-record factors is
//   the same number whichever record is the row, and one fused multiply-add replaces multiply, doubling and subtraction
//   (round 5; five float64 instructions per pair instead of seven).  `same`: the pair is a record with itself.
template <int METRIC>
__device__ __forceinline__ double gram_i8_value(const double G, const double t0r, const double t1r, const double* t2r,
                                                const double t0c, const double t1c, const double* t2c, const bool same) {
    if (METRIC == PO_EUCL) {
        const double sum = t0r + t0c;
        double d2 = fmax(fma(G, (-2.0 * t1r) * t1c, sum), 0.0);
        double scale = 1.0;
        if (d2 <= 0x1p-21 * sum) {
            // cancellation level.  (1) Are the two count vectors proportional, i.e. the frequency vectors identical (distance exactly 0 in
            // the reference)?  Cauchy-Schwarz equality G^2 == S_r S_c, tested exactly with error-free products (all three are integers
            // below 2^53).  (2) Otherwise the three rounded terms above have lost their leading bits to each other - at d^2 = 1e-11 sum the
            // distance is good to 2e-6 only, found by a fuzz seed with two near-identical records at the end of round 5 - and the squared
            // distance is taken from the integers instead:  d^2 = N / (n_r n_c)^2,  N = S_r n_c^2 + S_c n_r^2 - 2 G n_r n_c  = |n_c a - n_r b|^2,
            // in double-double arithmetic (error-free products and sums; every factor an exact integer, the totals recovered from their
            // rounded reciprocals), so that N keeps ~50 bits after a cancellation of 2^-50.  Symmetric in the two records like the form above.
            const double sr = *t2r, sc = *t2c;
            const double p = G * G, pe = fma(G, G, -p), q = sr * sc, qe = fma(sr, sc, -q);
            if (p == q && pe == qe) {
                d2 = 0.0;
            } else if (t1r > 0.0 && t1c > 0.0) {
                // (one product at a time, so that few values are live at once: the two-plane kernels run this next to 96 accumulators)
                double nc = __builtin_amdgcn_rcp(t1c);                            // totals < 2^32: reciprocal of the rounded reciprocal,
                nc = rint(fma(fma(-t1c, nc, 1.0), nc, nc));                       // one Newton step, rounds back to the integer
                double h = nc * nc, l = fma(nc, nc, -h);                          // n_c^2 < 2^64, exact as a pair
                const double ah = sr * h, al = fma(sr, h, -ah) + sr * l;          // S_r n_c^2
                double nr = __builtin_amdgcn_rcp(t1r);
                nr = rint(fma(fma(-t1r, nr, 1.0), nr, nr));
                h = nr * nr; l = fma(nr, nr, -h);
                const double bh = sc * h, bl = fma(sc, h, -bh) + sc * l;          // S_c n_r^2
                const double s1 = ah + bh, v1 = s1 - ah, e1 = ((ah - (s1 - v1)) + (bh - v1)) + (al + bl);
                h = nr * nc; l = fma(nr, nc, -h);
                const double g2 = 2.0 * G, ch = g2 * h, cl = fma(g2, h, -ch) + g2 * l;   // 2 G n_r n_c
                const double s2 = s1 - ch, v2 = s2 - s1, e2 = ((s1 - (s2 - v2)) + (-ch - v2)) + (e1 - cl);
                d2 = fmax(s2 + e2, 0.0);
                scale = t1r * t1c;                                                // 1 / (n_r n_c): d = sqrt(N) / (n_r n_c)
            }
        }
        const double v = po_sqrt_nonneg(d2) * scale;
        return same ? 0.0 : v;
    } else {                                               // SC; a constant record has N = 0 -> NaN as SciPy gives
        // 1 - G / sqrt(N_r N_c) as 1 - G (s_r s_c) with s = 1 / sqrt(N) once per record (t1; end of round 5: the per-pair v_rsq_f64 and its
        // two Newton steps were 10 of ~19 float64-rate instructions of a pair); the product of the two per-record factors is formed first, so
        // the value does not depend on which record is the row; identical records (G = N_r = N_c, exact integers) give exactly 0
        const double x = t0r * t0c;
        return (G == t0r && G == t0c) ? (x > 0.0 ? 0.0 : G / x) : fma(-G, t1r * t1c, 1.0);
    }
}

// The same values without their rare cases, N pairs at a time and step by step across the N (no branch, no load).  The result is
// the OR of the wave masks of the lanes where gram_i8_value has to be asked instead - a squared distance at cancellation level
// (or 0, or rounded below 0: the square root below has neither a clamp nor a zero guard), identical rank vectors - kept in scalar
// registers: it costs the vector ALU one comparison per pair.  A branch per pair keeps every pair's ~20 dependent float64
// instructions in a basic block of their own; written like this, N independent chains are in flight in one wave, which is what
// two waves per SIMD need to keep the vector ALU busy.  Where a lane's bit is clear its result equals gram_i8_value's bit for bit.
//   Eucl: gram_i8_value's test is  max(x, 0) <= 2^-21 sum.  Here (end of round 5: the clamp, the product and the float64 comparison
//   were 3 of ~17 float64-rate instructions per pair - 29.7 -> 27.7 ms at 200 000 records without them) the HIGH WORDS are compared:
//   for doubles 0 <= a <= b the high words are ordered the same way as integers, and so are they read as float32 (the patterns are
//   far from that format's NaNs), a negative x reads as a negative float32, so
//       hi(x) <=_f32 hi(sum) - (20 << 20)
//   holds for every pair the exact test holds for and a few more just above it (sum is 0 or >= 2^-64: S >= 1, totals < 2^32); the
//   subtraction saturates at 0, which is the case sum = 0 = x of two empty records.
template <int METRIC, int N, typename OUT>
__device__ __forceinline__ unsigned long long gram_i8_values_fast(const double (&G)[N], const double (&t0r)[N], const double (&t1r)[N],
                                                                  const double (&t0c)[N], const double (&t1c)[N], OUT (&out)[N]) {
    unsigned long long special = 0;
    double x[N], y[N];
    if (METRIC == PO_EUCL) {
        double gg[N], h[N], r[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const double sum = t0r[e] + t0c[e];
            x[e] = fma(G[e], (-2.0 * t1r[e]) * t1c[e], sum);
            uint32_t thr;
            asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(thr) : "v"((uint32_t)__double2hiint(sum)), "s"(20u << 20));
            special |= __builtin_amdgcn_ballot_w64(__uint_as_float((uint32_t)__double2hiint(x[e])) <= __uint_as_float(thr));
        }
#pragma unroll
        for (int e = 0; e < N; ++e) y[e] = __builtin_amdgcn_rsq(x[e]);      // po_sqrt_nonneg without its x == 0 case
#pragma unroll
        for (int e = 0; e < N; ++e) { gg[e] = x[e] * y[e]; h[e] = 0.5 * y[e]; }
#pragma unroll
        for (int e = 0; e < N; ++e) r[e] = fma(-h[e], gg[e], 0.5);
#pragma unroll
        for (int e = 0; e < N; ++e) { gg[e] = fma(gg[e], r[e], gg[e]); h[e] = fma(h[e], r[e], h[e]); }
#pragma unroll
        for (int e = 0; e < N; ++e) r[e] = fma(-gg[e], gg[e], x[e]);
#pragma unroll
        for (int e = 0; e < N; ++e) out[e] = (OUT)fma(r[e], h[e], gg[e]);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) special |= __builtin_amdgcn_ballot_w64(G[e] == t0r[e] && G[e] == t0c[e]);
#pragma unroll
        for (int e = 0; e < N; ++e) y[e] = t1r[e] * t1c[e];
#pragma unroll
        for (int e = 0; e < N; ++e) out[e] = (OUT)fma(-G[e], y[e], 1.0);
    }
    asm volatile("" : "+s"(special));                      // the comparisons are made here, not collected at the end of the caller
    return special;
}

// The values of a 32 x 64 block of a wave - records rrow .. rrow + 31 of the tile's rows (scratch rows trow ..), columns
// 64 wc .. - from its accumulators into the float32 tile scratch `tl` (po_tiles.h).  Accumulator layout (32x32): column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  kIlpRows rows
// = 2 kIlpRows pairs are in flight at a time (every pair holds ~10 registers); the rare cases follow wave by wave.
// terms: [t0 rows | t0 cols | t1 rows | t1 cols | t2 rows | t2 cols] of the tile's 128 + 128 records (LDS).
// REG0, NREG: the accumulator registers taken (all sixteen = the block's 32 rows; eight = sixteen of them, for the kernels that fill a
// scratch of fewer rows in several passes): register reg stands for row (reg & 3) + 8 ((reg - REG0) >> 2) + 4 (lane >> 5) behind rrow / trow.
template <int P, int METRIC, int kIlpRows, int STRIDE = kF32TileStride, typename OUT = float, int REG0 = 0, int NREG = 16>
__device__ __forceinline__ void gram_i8_values_to_tile(const v16i (&g)[2 * P - 1][2], const double* terms, const uint32_t rrow,
                                                       const uint32_t trow, const uint32_t wc, const uint32_t lr, const uint32_t lh,
                                                       const bool diag_tile, OUT* tl, const uint32_t ccol_of_block = 0xFFFFFFFFu) {
    constexpr int NG = 2 * P - 1;
    // the block's first column inside the tile (for the diagonal of the matrix): 64 wc, unless the scratch is a column half of the tile
    const uint32_t ccol = ccol_of_block == 0xFFFFFFFFu ? wc * 64 : ccol_of_block;
    const double* t0r = terms + rrow, *t0c = terms + 128 + wc * 64;             // Eucl: S/n^2      SC: N
    const double* t1r = terms + 256 + rrow, *t1c = terms + 384 + wc * 64;       // Eucl: 1/n        SC: 1/sqrt(N)
    const double* t2r = terms + 512 + rrow, *t2c = terms + 640 + wc * 64;       // Eucl: S
    OUT* wt = tl + (trow + 4 * lh) * STRIDE + wc * 64 + lr;
    // (two planes into a float64 scratch: 96 accumulators and value pairs of two registers each leave no room for the eight registers of
    //  column terms - they are read from the LDS again for every row, behind a lane index the compiler cannot see through)
    constexpr bool kReloadCols = sizeof(OUT) == 8 && P >= 2;
    double tc0 = t0c[lr], tc1 = t0c[32 + lr];
    double ic0 = t1c[lr], ic1 = t1c[32 + lr];                      // Eucl: 1/n      SC: 1/sqrt(N)
    auto gram = [&](const int nn, const int reg) -> double {
        double G = (double)g[NG - 1][nn][reg];         // Horner in 128: every partial sum an exact integer below 2^53
#pragma unroll
        for (int s = NG - 2; s >= 0; --s) G = fma(128.0, G, (double)g[s][nn][reg]);
        return G;
    };
    unsigned long long special = 0;                        // wave masks of the lanes with a rare case (scalar registers)
#pragma unroll
    for (int r0 = REG0; r0 < REG0 + NREG; r0 += kIlpRows) {
        double G[2 * kIlpRows], a0[2 * kIlpRows], a1[2 * kIlpRows], b0[2 * kIlpRows], b1[2 * kIlpRows];
        OUT v[2 * kIlpRows];
        if constexpr (kReloadCols) {
            uint32_t lrx = lr;
            asm volatile("" : "+v"(lrx));
            tc0 = t0c[lrx]; tc1 = t0c[32 + lrx];
            ic0 = t1c[lrx]; ic1 = t1c[32 + lrx];
        }
#pragma unroll
        for (int e = 0; e < kIlpRows; ++e) {
            const int reg = r0 + e;
            const uint32_t rl = (reg & 3) + 8 * ((reg - REG0) >> 2) + 4 * lh;
            G[2 * e] = gram(0, reg); G[2 * e + 1] = gram(1, reg);
            a0[2 * e] = a0[2 * e + 1] = t0r[rl];
            a1[2 * e] = a1[2 * e + 1] = t1r[rl];
            b0[2 * e] = tc0; b0[2 * e + 1] = tc1;
            b1[2 * e] = ic0; b1[2 * e + 1] = ic1;
        }
        special |= gram_i8_values_fast<METRIC, 2 * kIlpRows, OUT>(G, a0, a1, b0, b1, v);
        asm volatile("" : "+s"(special));                  // one running mask, not sixteen partial ones kept for a tree of ORs
#pragma unroll
        for (int e = 0; e < kIlpRows; ++e) {
            const int reg = r0 + e;
            OUT* wp = wt + ((reg & 3) + 8 * ((reg - REG0) >> 2)) * STRIDE;
            wp[0] = v[2 * e];
            wp[32] = v[2 * e + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the rare cases, wave by wave: a pair at cancellation level somewhere in the wave's block, or the diagonal of the matrix
    if (diag_tile || special != 0) {
        // (everything is derived again from the accumulators, behind a barrier the compiler cannot see through: values shared
        //  with the straight-line pass above would stay live across it - 32 Gram entries and 32 row terms are 128 registers)
        asm volatile("" ::: "memory");
        auto gram_again = [&](const int nn, const int reg) -> double {
            double G = 0.0;
#pragma unroll
            for (int s = NG - 1; s >= 0; --s) {
                int digit_sum = g[s][nn][reg];
                asm volatile("" : "+v"(digit_sum));
                G = s == NG - 1 ? (double)digit_sum : fma(128.0, G, (double)digit_sum);
            }
            return G;
        };
        // (two and three planes: all Gram entries of the block first - a pair of registers each instead of three or five accumulators -
        //  so that the exact branch of gram_i8_value has room next to them)
        double Gs[NG >= 3 ? NREG : 1][2];
        if constexpr (NG >= 3) {
#pragma unroll
            for (int reg = REG0; reg < REG0 + NREG; ++reg) {
                Gs[reg - REG0][0] = gram_again(0, reg);
                Gs[reg - REG0][1] = gram_again(1, reg);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the lane's coordinates once more, and every address from them: nothing per-lane is carried over from the pass above, and the
        //  column terms are read where they are used)
        uint32_t lane_c;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_c));
        const uint32_t lrc = lane_c & 31, lhc = lane_c >> 5;
        OUT* wtc = tl + (trow + 4 * lhc) * STRIDE + wc * 64 + lrc;
#pragma unroll
        for (int reg = REG0; reg < REG0 + NREG; ++reg) {
            const uint32_t rl = (reg & 3) + 8 * ((reg - REG0) >> 2) + 4 * lhc;
            OUT* wp = wtc + ((reg & 3) + 8 * ((reg - REG0) >> 2)) * STRIDE;
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                const uint32_t cl = 32 * nn + lrc;
                const double G = NG >= 3 ? Gs[NG >= 3 ? reg - REG0 : 0][nn] : gram_again(nn, reg);
                wp[32 * nn] = (OUT)gram_i8_value<METRIC>(G, t0r[rl], t1r[rl], t2r + rl, t0c[cl], t1c[cl], t2c + cl, diag_tile && rrow + rl == ccol + cl);
                __builtin_amdgcn_sched_barrier(0);         // (one value after the other: their exact branches are long)
            }
        }
    }
}

// P digit planes; runs iff  run_above < *maxabs <= run_upto  (maxabs == nullptr: always).
template <int P, int METRIC, typename OUT>
__global__ __launch_bounds__(kThreads, P == 1 ? 4 : 2) void gram_i8_tile_kernel(po_tile_args A, const int8_t* __restrict__ planes,
                                                                               uint32_t dpad, const double* __restrict__ rs,
                                                                               const uint32_t* __restrict__ maxabs,
                                                                               long long run_above, long long run_upto, tile_list tiles) {
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
    list_tile(A, tiles, blockIdx.x, ti, tj);
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
    // (lane coordinates derived again: anything per-lane that lives across the matrix-core loop is spilled at the register limit)
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t t_e = wave * 64u + lane_e, lr_e = lane_e & 31, lh_e = lane_e >> 5;
    constexpr bool F32 = sizeof(OUT) == 4;                 // float32 output: the whole tile goes through LDS (po_store_tile_f32)
    double* terms = reinterpret_cast<double*>(smem + (F32 ? kF32TileBytes : kMirrorBytes));   // [t0 rows | t0 cols | t1 rows | t1 cols]
    if (t_e < 256) {
        const uint64_t rec = (t_e < 128) ? i0 + t_e : j0 + (t_e - 128);
        terms[t_e] = rs[rec];
        terms[256 + t_e] = rs[A.npad + rec];
        terms[512 + t_e] = METRIC == PO_EUCL ? rs[2 * A.npad + rec] : 0.0;
    }
    __syncthreads();

    // ---- epilogue: accumulator layout (32x32): column = lane_e & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane_e >> 5)
    const bool mirror = po_tile_mirrors(A, ti, tj);
    const uint64_t iw = i0 + wr * 32, jw = j0 + wc * 64;
    const double* t0r = terms + wr * 32, *t0c = terms + 128 + wc * 64;          // Eucl: S/n^2      SC: N
    const double* t1r = terms + 256 + wr * 32, *t1c = terms + 384 + wc * 64;    // Eucl: 1/n
    const double* t2r = terms + 512 + wr * 32, *t2c = terms + 640 + wc * 64;    // Eucl: S
    // the distance of row record iw + rl and column record jw + cl from their exact integer Gram entry
    const bool diag_tile = ti == tj;
    auto value = [&](const double G, const uint32_t rl, const uint32_t cl, const double tc, const double ic) -> double {
        return gram_i8_value<METRIC>(G, t0r[rl], t1r[rl], t2r + rl, tc, ic, t2c + cl,
                                     diag_tile && wr * 32 + rl == wc * 64 + cl);
    };
    auto gram = [&](const int nn, const int reg) -> double {
        double G = (double)g[NG - 1][nn][reg];             // Horner in 128: every partial sum an exact integer below 2^53
#pragma unroll
        for (int s = NG - 2; s >= 0; --s) G = fma(128.0, G, (double)g[s][nn][reg]);
        return G;
    };
    if constexpr (F32) {
        // every value once into the tile-shaped LDS scratch (no global address, no bounds test per element), one barrier,
        // then 16-byte stores of whole 512-byte row pieces for the tile and for its transpose
        float* tl = reinterpret_cast<float*>(smem);
        gram_i8_values_to_tile<P, METRIC, 2>(g, terms, wr * 32, wr * 32, wc, lr_e, lh_e, diag_tile, tl);
        po_lds_barrier();
        po_store_tile_f32<kThreads / 64>(A, mirror, i0, j0, wave, lane_e, tl);
    } else {
        OUT* out = static_cast<OUT*>(A.out);
        OUT* mir = static_cast<OUT*>(A.mirror);
        const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
        double* wl = reinterpret_cast<double*>(smem) + wave * (32 * kTrStride);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const uint64_t c = jw + nn * 32 + lr_e;
            const double tc = t0c[nn * 32 + lr_e], ic = t1c[nn * 32 + lr_e];
            const bool c_ok = c >= A.col_begin && c < n_cols;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh_e;
                const uint64_t r = iw + rl;
                const double v = value(gram(nn, reg), rl, nn * 32 + lr_e, tc, ic);
                if (c_ok && r >= A.row_begin && r < n_rows) po_out_store(&out[(r - A.row_begin) * A.ld_out + (c - A.col_begin)], (OUT)v);
                if (mirror) wl[lr_e * kTrStride + rl] = v;
            }
            if (mirror) {                                  // wave-private scratch; LDS operations of a wave run in order
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const uint32_t jr = it * 2 + lh_e;
                    const double w = wl[jr * kTrStride + lr_e];
                    const uint64_t cm = jw + nn * 32 + jr, r = iw + lr_e;
                    if (cm >= A.col_begin && cm < n_cols && r >= A.row_begin && r < n_rows)
                        po_out_store(&mir[(cm - A.col_begin) * A.ld_mirror + (r - A.row_begin)], (OUT)w);
                }
            }
        }
    }
}

// ---- float32 output, one digit plane: four waves of 64 x 64 pairs, four workgroups per CU (round 5) ---------------------
// The phases of a tile - operands in, matrix cores, ~20 float64 instructions per pair, scratch, stores out - use different
// parts of the CU and follow each other inside a workgroup; what overlaps them is OTHER workgroups of the CU.  The eight-wave
// kernel above has two per CU (72 KiB of LDS each).  This one has four: 256 lanes, a wave holds 2 x 2 accumulator blocks (64
// registers; a third less LDS operand traffic per matrix instruction), and the float32 scratch holds HALF a tile (33 KiB): row
// block bi of wave row wr covers the tile's rows 64 bi + 32 wr .., so "block 0 of every wave" is the tile's upper half and
// "block 1" the lower one, and the epilogue makes two passes - values of one half into the scratch, barrier, stores
// (po_store_tile_f32<4, 64>), barrier.  39 KiB of LDS, at most 128 registers.
constexpr int kQuadThreads = 256;
constexpr int kQuadScratchBytes = 64 * kF32TileStride * 4;             // 33 280 B (the 32 KiB staging area lies underneath)
constexpr int kQuadLdsBytes = kQuadScratchBytes + kTermBytes;

template <int METRIC, typename OUT>
__global__ __launch_bounds__(kQuadThreads, 4) void gram_i8_quad_kernel(po_tile_args A, const int8_t* __restrict__ planes, uint32_t dpad,
                                                                       const double* __restrict__ rs, const uint32_t* __restrict__ maxabs,
                                                                       long long run_above, long long run_upto, tile_list tiles) {
    if (maxabs != nullptr) {
        const long long m = *maxabs;
        if (m <= run_above || m > run_upto) return;
    }
    extern __shared__ __align__(16) unsigned char smem[];  // staging [A|B][8 chunks][128 records][16 B], then the scratch; terms behind
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t wr = wave >> 1, wc = wave & 1;          // 2 x 2 waves of 64 x 64
    const uint32_t lr = lane & 31, lh = lane >> 5;
    uint32_t ti, tj;
    list_tile(A, tiles, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;

    // the per-record terms of the epilogue come by LDS-DMA with the first staging step (their part of the LDS lies behind the
    // scratch, outside the staging area): a global load behind the matrix-core loop would be waited for in the open
    double* terms = reinterpret_cast<double*>(smem + kQuadScratchBytes);    // [t0 rows | t0 cols | t1 rows | t1 cols | t2 rows | t2 cols]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t idx = wave * 2 + u, a = idx >> 1, side = idx & 1;   // six (Eucl) or two (SC) one-KiB instructions
        if (idx < (METRIC == PO_EUCL ? 6u : 4u))
            po_glds16(rs + a * A.npad + (side ? j0 : i0) + 2 * lane, reinterpret_cast<unsigned char*>(terms + a * 256 + side * 128));
    }

    // Which tile row a lane's operand row stands for.  float32: row block bi of wave row wr = tile rows 64 bi + 32 wr + m (the scratch holds
    // 64 rows: two passes, "block bi of every wave").  float64: the same 33 KiB hold 32 rows of doubles, so the epilogue makes FOUR passes
    // and a pass must be 32 consecutive tile rows with every wave contributing: m = 16 hi + m' of wave row wr = tile row 64 bi + 32 hi +
    // 16 wr + m' - registers 8 hi .. 8 hi + 7 of block bi are pass 2 bi + hi.
    constexpr bool F64 = sizeof(OUT) == 8;
    const uint32_t arow = F64 ? 32 * (lr >> 4) + 16 * wr + (lr & 15) : wr * 32 + lr;

    v16i g[2][1][2];                                       // [row block bi][plane sum][column block]
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[bi][0][nn][e] = 0;

    for (uint32_t k0 = 0; k0 < dpad; k0 += KCH) {
        __syncthreads();                                   // the previous step's operand reads are done
#pragma unroll
        for (int u = 0; u < 8; ++u) {                      // 32 one-KiB LDS-DMA instructions, 8 per wave
            const uint32_t idx = wave * 8 + u;
            const uint32_t side = (idx >> 4) & 1, q = (idx >> 1) & 7, half = idx & 1;
            const uint64_t rec = (side ? j0 : i0) + half * 64 + lane;
            po_glds16(planes + ((size_t)(k0 / 16 + q) * A.npad + rec) * 16, smem + (side * 8 + q) * kChunkBytes + half * 1024);
        }
        __syncthreads();                                   // drains the LDS-DMA (vmcnt) of every wave
#pragma unroll
        for (int s = 0; s < KCH / 32; ++s) {
            const uint32_t q = 2 * s + lh;                 // lane halves take the two 16-byte chunks of a k-step
            v4i a[2], b[2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) a[bi] = *reinterpret_cast<const v4i*>(smem + q * kChunkBytes + (bi * 64 + arow) * 16);
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) b[nn] = *reinterpret_cast<const v4i*>(smem + (8 + q) * kChunkBytes + (wc * 64 + nn * 32 + lr) * 16);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) g[bi][0][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[bi], b[nn], g[bi][0][nn], 0, 0, 0);
        }
    }
    __syncthreads();                                       // the staging area becomes the scratch
    // (lane coordinates derived again: anything per-lane that lives across the matrix-core loop is spilled at the register limit)
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t lr_e = lane_e & 31, lh_e = lane_e >> 5;
    OUT* tl = reinterpret_cast<OUT*>(smem);
    const bool mirror = po_tile_mirrors(A, ti, tj);
    constexpr int kPasses = F64 ? 4 : 2, kPassRows = 128 / kPasses;
    auto pass = [&](auto H) {                              // (the pass number is a template argument of the value function)
        constexpr int h = decltype(H)::value;
        po_lds_barrier();                                  // terms in place (h = 0); the stores of the pass before have read the scratch
        if constexpr (F64)
            gram_i8_values_to_tile<1, METRIC, 2, kF32TileStride, double, 8 * (h & 1), 8>(g[h >> 1], terms, h * 32 + wr * 16, wr * 16, wc, lr_e, lh_e,
                                                                                     ti == tj, tl);
        else
            gram_i8_values_to_tile<1, METRIC, 2>(g[h], terms, h * 64 + wr * 32, wr * 32, wc, lr_e, lh_e, ti == tj, tl);
        po_lds_barrier();
        // (the lane index once more, per pass: store addresses shared between the passes would stay live across the next pass's
        //  arithmetic - next to its accumulators - and spill)
        uint32_t lane_s;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
        po_store_tile<OUT, kQuadThreads / 64, kPassRows, 128, kF32TileStride>(A, mirror, i0 + h * kPassRows, j0, wave, lane_s, tl);
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    if constexpr (F64) {
        pass(std::integral_constant<int, 2>{});
        pass(std::integral_constant<int, 3>{});
    }
}

// ---- float32 output, two digit planes: four waves on HALF a tile (128 x 64 pairs), four workgroups per CU (round 5) --------
// Two planes are three accumulator groups: 96 registers for a 32 x 64 wave block, which rules out the 64 x 64 blocks of the quad
// kernel at four workgroups per CU.  Here a workgroup of four waves takes the left or the right half of a tile - wave w rows
// 32 w .., all 64 columns of the half -, the operands arrive in steps of 64 bytes of K (24 KiB), the scratch holds the half
// (128 x 64 float32, row stride 66: its rows leave as 256-byte pieces, its transposed rows as 512-byte pieces): 39 KiB of LDS, <= 128
// registers, four workgroups per CU.  The two halves of a tile run on the same XCD one after the other (workgroup b: XCD b % 8, half
// (b / 8) & 1), so the row operand they share is read from HBM once.
constexpr int kHalfThreads = 256;
constexpr int kHalfStride = 66;
constexpr int kHalfKch = 64;                                           // bytes of K per record and staging step
constexpr int kHalfStageA = 2 * (kHalfKch / 16) * kChunkBytes;         // [p][4 chunks][128 records][16 B] = 16 KiB
constexpr int kHalfStageBytes = kHalfStageA + kHalfStageA / 2;         // + the same for the 64 column records
constexpr int kHalfScratchBytes = 128 * kHalfStride * 4;               // 33 792 B, over the staging area
constexpr int kHalfLdsBytes = kHalfScratchBytes + kTermBytes;

template <int METRIC, typename OUT>
__global__ __launch_bounds__(kHalfThreads, 4) void gram_i8_half_kernel(po_tile_args A, const int8_t* __restrict__ planes, uint32_t dpad,
                                                                       const double* __restrict__ rs, const uint32_t* __restrict__ maxabs,
                                                                       long long run_above, long long run_upto, tile_list tiles) {
    static_assert(kHalfStageBytes <= kHalfScratchBytes, "the scratch covers the staging area");
    if (maxabs != nullptr) {
        const long long m = *maxabs;
        if (m <= run_above || m > run_upto) return;
    }
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t lr = lane & 31, lh = lane >> 5;
    const uint32_t xcd = blockIdx.x % kXcds, v = blockIdx.x / kXcds;
    const uint32_t hcol = v & 1;                                       // which half of the tile's columns
    const uint64_t w = (uint64_t)(v >> 1) * kXcds + xcd;               // the tile's virtual workgroup index (w % 8 = this XCD)
    if (w >= tiles.count) return;                                      // (uniform: the grid is rounded up to whole groups of 8)
    uint32_t ti, tj;
    list_tile(A, tiles, w, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN + hcol * 64;
    const size_t plane = (size_t)A.npad * dpad;

    // float32: wave w holds tile rows 32 w + m, the scratch the whole half (128 x 64).  float64: the same 33 KiB hold 64 rows of doubles -
    // two passes, each 64 consecutive tile rows with every wave contributing: m = 16 hi + m' of wave w = tile row 64 hi + 16 w + m'.
    constexpr bool F64 = sizeof(OUT) == 8;
    const uint32_t arow = F64 ? 64 * (lr >> 4) + 16 * wave + (lr & 15) : wave * 32 + lr;
    v16i g[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[s][nn][e] = 0;

    for (uint32_t k0 = 0; k0 < dpad; k0 += kHalfKch) {
        __syncthreads();                                   // the previous step's operand reads are done
#pragma unroll
        for (int u = 0; u < 6; ++u) {                      // 24 one-KiB LDS-DMA instructions, 6 per wave: 16 for the rows, 8 for the columns
            const uint32_t idx = wave * 6 + u;
            if (idx < 16) {
                const uint32_t p = idx >> 3, q = (idx >> 1) & 3, half = idx & 1;
                po_glds16(planes + p * plane + ((size_t)(k0 / 16 + q) * A.npad + i0 + half * 64 + lane) * 16,
                          smem + (p * 4 + q) * kChunkBytes + half * 1024);
            } else {
                const uint32_t p = (idx - 16) >> 2, q = (idx - 16) & 3;
                po_glds16(planes + p * plane + ((size_t)(k0 / 16 + q) * A.npad + j0 + lane) * 16,
                          smem + kHalfStageA + (p * 4 + q) * (kChunkBytes / 2));
            }
        }
        __syncthreads();                                   // drains the LDS-DMA (vmcnt) of every wave
#pragma unroll
        for (int s = 0; s < kHalfKch / 32; ++s) {
            const uint32_t q = 2 * s + lh;                 // lane halves take the two 16-byte chunks of a k-step
            v4i a[2], b[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a[p] = *reinterpret_cast<const v4i*>(smem + (p * 4 + q) * kChunkBytes + arow * 16);
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    b[p][nn] = *reinterpret_cast<const v4i*>(smem + kHalfStageA + (p * 4 + q) * (kChunkBytes / 2) + (nn * 32 + lr) * 16);
            }
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
                        g[pa + pb][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[pa], b[pb][nn], g[pa + pb][nn], 0, 0, 0);
        }
    }
    __syncthreads();                                       // the staging area becomes the scratch
    // (lane coordinates derived again: anything per-lane that lives across the matrix-core loop is spilled at the register limit)
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t t_e = wave * 64u + lane_e, lr_e = lane_e & 31, lh_e = lane_e >> 5;
    OUT* tl = reinterpret_cast<OUT*>(smem);
    double* terms = reinterpret_cast<double*>(smem + kHalfScratchBytes);      // [t0 rows 128 | t0 cols (64 used) | t1 .. | t2 ..]
    if (t_e < 192) {
        const uint64_t rec = (t_e < 128) ? i0 + t_e : j0 + (t_e - 128);
        terms[t_e] = rs[rec];
        terms[256 + t_e] = rs[A.npad + rec];
        terms[512 + t_e] = METRIC == PO_EUCL ? rs[2 * A.npad + rec] : 0.0;
    }
    __syncthreads();
    const bool mirror = po_tile_mirrors(A, ti, tj);
    if constexpr (!F64) {
        gram_i8_values_to_tile<2, METRIC, 1, kHalfStride>(g, terms, wave * 32, wave * 32, 0, lr_e, lh_e, ti == tj, tl, hcol * 64);
        po_lds_barrier();
        uint32_t lane_s;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
        po_store_tile<float, kHalfThreads / 64, 128, 64, kHalfStride>(A, mirror, i0, j0, wave, lane_s, tl);
    } else {
        auto pass = [&](auto H) {
            constexpr int h = decltype(H)::value;
            if (h) po_lds_barrier();                       // the stores of the upper rows have read the scratch
            // (lane coordinates per pass: LDS addresses shared between the passes would live across the first one's arithmetic and spill)
            uint32_t lane_p;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
            gram_i8_values_to_tile<2, METRIC, 1, kHalfStride, double, 8 * h, 8>(g, terms, h * 64 + wave * 16, wave * 16, 0, lane_p & 31, lane_p >> 5,
                                                                                ti == tj, tl, hcol * 64);
            po_lds_barrier();
            uint32_t lane_s;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
            po_store_tile<double, kHalfThreads / 64, 64, 64, kHalfStride>(A, mirror, i0 + h * 64, j0, wave, lane_s, tl);
        };
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{});
    }
}

// (A persistent variant - one workgroup per CU, eight computing waves that never wait for a store, a ninth wave staging operands and
//  terms by LDS-DMA, hand-over by s_barrier only - was built first in round 5 and is not in the tree: with ONE workgroup per CU the phases
//  of a tile add up exactly (19.1 ms skeleton + 8.7 arithmetic + 9.0 stores = 37.4 ms at 200 000 records), because a store instruction is
//  asynchronous only as deep as the memory pipeline's queues.  profiles/r05_float32_output.txt section 3; git history: 736e647.)

// `count` tiles: all of the block's (d_list == nullptr) or the listed ones.  The kernel runs iff run_above < *maxabs <= run_upto.
template <int P, int METRIC>
int launch_tiles(po_ctx* ctx, const po_tile_args& a, const int8_t* planes, uint32_t dpad, const double* rs,
                 const uint32_t* maxabs, long long run_above, long long run_upto, const uint32_t* d_list, uint64_t count) {
    if (count == 0) return PO_OK;
    if (count >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)count); return PO_EUNSUPPORTED; }
    const tile_list tl{d_list, count};
    const size_t staging = (size_t)P * 2 * 8 * kChunkBytes;
    const size_t scratch = a.out_f32 ? (size_t)kF32TileBytes : (size_t)kMirrorBytes;      // the epilogue's LDS, over the staging area
    const size_t shmem = (staging > scratch ? staging : scratch) + kTermBytes;
    // one plane: four waves of 64 x 64 pairs; two planes: four waves on half a tile (at every width: k = 5 / 6 Spearman in float32 6.1 /
    // 3.3 ms against 7.0 / 3.7 for the eight-wave kernel on whole tiles) - four workgroups per CU, either output type
    if constexpr (P == 1) {
        if (a.out_f32) {
            auto k = gram_i8_quad_kernel<METRIC, float>;
            PO_SHMEM(ctx, k, (size_t)kQuadLdsBytes);
            hipLaunchKernelGGL(k, dim3((uint32_t)count), dim3(kQuadThreads), (size_t)kQuadLdsBytes, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
        } else {
            auto k = gram_i8_quad_kernel<METRIC, double>;
            PO_SHMEM(ctx, k, (size_t)kQuadLdsBytes);
            hipLaunchKernelGGL(k, dim3((uint32_t)count), dim3(kQuadThreads), (size_t)kQuadLdsBytes, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
        }
        PO_CHECK_LAUNCH("gram_i8_quad_kernel");
        return PO_OK;
    }
    // (float64 from 2 048 words on - k = 6 Spearman, matrix-core bound - is the one case left to the eight-wave kernel: 3.26 against 3.46 ms
    //  at 20 000 records; k = 5: half tiles 6.75 against 6.9)
    if constexpr (P == 2) if (a.out_f32 || dpad < 2048) {
        const uint64_t grid = 2 * po_round_up(count, kXcds);           // both halves of a tile on the tile's XCD
        if (a.out_f32) {
            auto k = gram_i8_half_kernel<METRIC, float>;
            PO_SHMEM(ctx, k, (size_t)kHalfLdsBytes);
            hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kHalfThreads), (size_t)kHalfLdsBytes, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
        } else {
            auto k = gram_i8_half_kernel<METRIC, double>;
            PO_SHMEM(ctx, k, (size_t)kHalfLdsBytes);
            hipLaunchKernelGGL(k, dim3((uint32_t)grid), dim3(kHalfThreads), (size_t)kHalfLdsBytes, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
        }
        PO_CHECK_LAUNCH("gram_i8_half_kernel");
        return PO_OK;
    }
    if constexpr (P >= 2) {                                // three planes: eight waves on whole tiles (160 accumulator registers)
        if (a.out_f32) {
            if constexpr (P == 3) {
                auto k = gram_i8_tile_kernel<P, METRIC, float>;
                PO_SHMEM(ctx, k, shmem);
                hipLaunchKernelGGL(k, dim3((uint32_t)count), dim3(kThreads), shmem, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
            }
        } else {
            auto k = gram_i8_tile_kernel<P, METRIC, double>;
            PO_SHMEM(ctx, k, shmem);
            hipLaunchKernelGGL(k, dim3((uint32_t)count), dim3(kThreads), shmem, ctx->stream, a, planes, dpad, rs, maxabs, run_above, run_upto, tl);
        }
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

// The largest counts on the host: [whole matrix, every block of 128 records].  Waits for the stream (16 bytes + 4 per block).
int po_gram_i8_block_maxima(po_ctx* ctx, const void* ws, uint64_t npad, uint32_t dim, const uint32_t** h_max) {
    const ws_view v = view(const_cast<void*>(ws), npad, dim);
    const size_t words = 1 + npad / 128;
    if (ctx->h_blockmax_cap < words) {
        if (ctx->h_blockmax) (void)hipHostFree(ctx->h_blockmax);
        ctx->h_blockmax = nullptr; ctx->h_blockmax_cap = 0;
        const size_t cap = words + words / 2 + 64;
        PO_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_blockmax), cap * sizeof(uint32_t), hipHostMallocDefault));
        ctx->h_blockmax_cap = cap;
    }
    PO_HIP(hipMemcpyAsync(ctx->h_blockmax, v.maxabs, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    *h_max = ctx->h_blockmax;
    return PO_OK;
}

// Eucl.  Without the host's copy of the largest counts (h_max == NULL: small matrices, where a host synchronisation would cost more
// than workgroups that leave at once): the one-, two- and three-plane kernels (max <= 127, <= 16 383, <= limit) are all launched over
// all tiles and one of them runs.  With it (round 5): a tile needs as many digit planes as the larger of its two record blocks'
// largest counts - in a real assembly a handful of Mb-scale contigs need the third digit, most blocks one or two - so the tiles
// are dealt to the kernels by class: the host counts the tiles of every class from the block maxima (products of block counts),
// classify_tiles_kernel writes their positions, and every kernel is launched over exactly its own list.  One class only (every
// BASELINE config): no lists, one launch.  Results do not depend on the class a tile runs in (the Gram entries are exact).
// SC: two planes, unconditionally.
int po_launch_gram_i8_tiles(po_ctx* ctx, int metric, const po_tile_args& a, const void* ws, const uint32_t* h_max, uint64_t* tiles) {
    const ws_view v = view(const_cast<void*>(ws), a.npad, a.dim);
    const uint64_t nb = po_tile_count(a, TM);
    if (tiles) *tiles += nb;
    if (nb == 0) return PO_OK;
    constexpr long long kAlways = 0x7fffffffffffffffll;
    if (metric == PO_SC) return launch_tiles<2, PO_SC>(ctx, a, v.planes, v.dpad, v.rs, nullptr, -1, kAlways, nullptr, nb);
    if (metric != PO_EUCL) { po_set_error("po_launch_gram_i8_tiles: metric %d is not a Gram-form metric", metric); return PO_EINVAL; }
    const uint32_t limit = po_gram_i8_value_limit(a.dim);
    if (h_max == nullptr) {
        int rc = launch_tiles<1, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, -1, 127, nullptr, nb);
        if (rc) return rc;
        if (limit > 127) rc = launch_tiles<2, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, 127, limit < 16383u ? limit : 16383u, nullptr, nb);
        if (rc) return rc;
        if (limit > 16383) rc = launch_tiles<3, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, 16383, limit, nullptr, nb);
        return rc;
    }
    if (h_max[0] > limit) return PO_OK;                   // the float64 Gram kernel owns the matrix
    // tiles of class <= c: (row blocks of class <= c) x (column blocks of class <= c); in a triangular block m (m + 1) / 2
    auto cls = [&](uint64_t blk) { const uint32_t m = h_max[1 + blk]; return m <= 127u ? 0 : (m <= 16383u ? 1 : 2); };
    const uint64_t r0 = a.row_begin / TM, r1 = (a.row_end + TM - 1) / TM, c0 = a.col_begin / TN, c1 = (a.col_end + TN - 1) / TN;
    uint64_t rows_le[3] = {0, 0, 0}, cols_le[3] = {0, 0, 0}, le[3];
    for (uint64_t b = r0; b < r1; ++b) for (int c = cls(b); c < 3; ++c) ++rows_le[c];
    for (uint64_t b = c0; b < c1; ++b) for (int c = cls(b); c < 3; ++c) ++cols_le[c];
    for (int c = 0; c < 3; ++c) le[c] = a.triangular ? rows_le[c] * (rows_le[c] + 1) / 2 : rows_le[c] * cols_le[c];
    const uint64_t count[3] = {le[0], le[1] - le[0], le[2] - le[1]};
    if (le[2] != nb) { po_set_error("po_launch_gram_i8_tiles: %llu tiles classified, %llu in the block", (unsigned long long)le[2], (unsigned long long)nb); return PO_EINVAL; }
    const uint32_t* lists = nullptr;
    if ((count[0] != 0) + (count[1] != 0) + (count[2] != 0) > 1) {
        if (nb >= (1ull << 32)) { po_set_error("too many tiles for a tile list (%llu)", (unsigned long long)nb); return PO_EUNSUPPORTED; }
        int rc = po_buf_reserve(ctx, &ctx->ws_tilelist, 3 * nb * sizeof(uint32_t) + 64);
        if (rc) return rc;
        uint32_t* counters = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(ctx->ws_tilelist.p) + 3 * nb * sizeof(uint32_t));
        PO_HIP(hipMemsetAsync(counters, 0, 16, ctx->stream));
        hipLaunchKernelGGL(classify_tiles_kernel, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, ctx->stream, a, v.maxabs, nb,
                           static_cast<uint32_t*>(ctx->ws_tilelist.p), counters);
        PO_CHECK_LAUNCH("classify_tiles_kernel");
        lists = static_cast<const uint32_t*>(ctx->ws_tilelist.p);
    }
    int rc = launch_tiles<1, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, -1, kAlways, lists, count[0]);
    if (rc) return rc;
    rc = launch_tiles<2, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, -1, kAlways, lists ? lists + nb : nullptr, count[1]);
    if (rc) return rc;
    return launch_tiles<3, PO_EUCL>(ctx, a, v.planes, v.dpad, v.rs, v.maxabs, -1, kAlways, lists ? lists + 2 * nb : nullptr, count[2]);
}
