// Stage 2, JSD fast path for record pairs with EQUAL word totals: integer-sum table lookup.
//
// Same mathematics as valu_tile_kernel<JSD> (phylodist.JSD / KL,
// /root/reference/phylopackage/core/phylodist.py:18-24, :43-48).  When two records have the same number
// of counted words n (fixed-length contigs, sliding windows, simulated reads), the mixture
// h = (a+b)/2 has h_w = (ca_w + cb_w) / 2n, so
//     S = sum_w (a_w+b_w) ln(a_w+b_w) = (1/n) sum_w T[ca_w + cb_w] - 2 ln n,   T[x] = x ln x,
// and the per-word, per-pair work collapses from a float64 logarithm to ONE integer add, ONE LDS read
// and ONE float64 add.  T[x] for x = 0..255 sits in LDS replicated 32x (entry x, copy c at byte
// x*256 + c*8) so that the per-lane lookups of a wave never conflict; counts are kept pre-shifted by 8
// bits (and the column side with its lane's copy offset baked in), so a lookup address is a single v_add_u32.
//
// Eligibility is decided per tile on the device: classify_kernel marks each block of 128 records with its
// common total (0 = mixed / empty / a count above 127); a tile (I,J) takes this path iff both classes are
// equal and non-zero, every other tile is left to valu_tile_kernel<JSD>, which skips the marked ones.
#include "po_tiles.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int TM = 128, TN = 128;
constexpr int kLutEntries = 256;                       // sums 0..255  -> counts up to 127 (fixed-length contigs up to ~10 kb at k=4)
constexpr int kLutBytes = kLutEntries * 256;           // 32 copies x 8 B per entry
// Round 4, the WIDE layout of the same 64 KiB: 512 entries x 16 copies (sums 0..511 -> counts up to 255: fixed-length records
// of 10 .. 20 kb at k=4, e.g. a genome cut into windows).  Lanes l and l + 16 of a 32-lane LDS pass then share a copy (a 2-way
// bank conflict when they read different entries), which is still several times cheaper than the float64-logarithm kernel such an
// input used to fall back to.  The layout is chosen on the device from the largest count: prep_counts_kernel<false> writes the
// operands for the narrow layout and finds the maximum, prep_counts_kernel<true> rewrites them for the wide one iff
// 127 < max <= 255 (it leaves at once otherwise), the table is laid out accordingly; the tile kernel only ever adds a row term
// and a column term and is the same for both.
constexpr uint32_t kNarrowMax = 127, kWideMax = 255;
constexpr double LN2 = 0.693147180559945309417232121458;

// Ct[d][npad] = counts[n][d] << 8 (uint32, transposed, zero padded to D8 x npad)
// ctb (may be NULL): the same with the byte offset of the table copy of the lane that will look the column up baked in:
// column record j is looked up by lane (j >> 1) & 63 of a wave, which uses copy (j >> 1) & 31 (8 bytes each)
template <bool WIDE>
__global__ __launch_bounds__(256) void prep_counts_kernel(const uint32_t* __restrict__ counts, uint64_t n, uint32_t dim,
                                                          uint64_t npad, uint32_t* __restrict__ ct, uint32_t* __restrict__ ctb,
                                                          uint32_t* __restrict__ maxcount) {
    if (WIDE) {                                                   // second pass: only when the first one found 127 < max <= 255
        const uint32_t m = *maxcount;
        if (m <= kNarrowMax || m > kWideMax) return;
    }
    constexpr uint32_t SHIFT = WIDE ? 7u : 8u, COPY_MASK = WIDE ? 15u : 31u;
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
        tile[r][tx] = v << SHIFT;
    }
    if (!WIDE) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&blkmax, mx);
    }
    __syncthreads();
    if (!WIDE && threadIdx.x == 0 && blkmax > *maxcount) atomicMax(maxcount, blkmax);   // racy pre-check skips redundant atomics
    for (uint32_t r = ty; r < 64; r += 4)
        if (d0 + r < ((dim + 7u) & ~7u) && n0 + tx < npad) {
            ct[(uint64_t)(d0 + r) * npad + n0 + tx] = tile[tx][r];
            if (ctb) ctb[(uint64_t)(d0 + r) * npad + n0 + tx] = tile[tx][r] + ((((uint32_t)(n0 + tx) >> 1) & COPY_MASK) << 3);
        }
}

// cls[b] = common word total of records [128b, 128b+128) (padding ignored), 0 if they differ, if one is
// empty, or if any count in the matrix exceeds what the table covers.
// The table identity needs sum_w count = total for every record (what count2freq guarantees); a caller-supplied
// total that disagrees with its counts shows up as wsum = sum_w count/total != 1 and sends the block to the general kernel.
// clsk[b] = {1/n, 2 ln n} of that total: the two constants of the tile kernel's epilogue, computed once per block of records here
// instead of once per wave and tile there (a float64 logarithm and a division are ~200 vector instructions).
__global__ __launch_bounds__(128) void classify_kernel(const unsigned long long* __restrict__ totals, uint64_t n,
                                                       const uint32_t* __restrict__ maxcount,
                                                       const double* __restrict__ wsum,
                                                       unsigned long long* __restrict__ cls, double2* __restrict__ clsk) {
    const uint64_t r = (uint64_t)blockIdx.x * 128 + threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * 128;
    const unsigned long long ref = totals[first];                     // first < n by construction of the grid
    const bool ok = (r >= n) || (totals[r] == ref && fabs(wsum[r] - 1.0) < 1e-9);
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) {
        const bool table = all && ref > 0 && *maxcount <= kWideMax;
        cls[blockIdx.x] = table ? ref : 0ull;
        clsk[blockIdx.x] = table ? make_double2(1.0 / (double)ref, 2.0 * log((double)ref)) : make_double2(0.0, 0.0);
    }
}

// the table as the tile kernels want it in LDS: entry x replicated 32 times (copy c at x*32 + c) - or, in the wide layout,
// 512 entries replicated 16 times (copy c at x*16 + c): the same 8 192 doubles either way
__global__ void lut_table_kernel(double* __restrict__ lut, const uint32_t* __restrict__ maxcount) {
    const uint32_t g = blockIdx.x * 32 + threadIdx.x;
    const uint32_t x = *maxcount > kNarrowMax ? g >> 4 : g >> 5;
    lut[g] = x ? (double)x * log((double)x) : 0.0;
}



// ---- the tile kernel: wave-uniform rows (round 3) ------------------------------------------------------------------
// A wave owns RW rows x 128 columns of the tile, a lane two adjacent columns of it.  The RW row counts of a word are the
// same for every lane: they are fetched by ONE scalar load (s_load_dwordx8 / x16 from the transposed count matrix) and sit in
// scalar registers, so that a lookup address is one v_add_u32 in its short encoding (scalar row term + vector column term) and
// nothing else; the two column counts of a word come straight from global memory (8 bytes per lane, 512 contiguous bytes per
// wave, requested kPF words ahead) with the lane's table copy baked in (ctb).  No operand goes through LDS any more - no
// LDS-DMA staging, no barrier in the loop - and the LDS data path serves nothing but the lookups (the round-1/2 kernel, a
// 4 x 8 register block per lane with both operands staged through LDS, spent 1 ds_read_b128 + 4 ds_read_b64 of every 84 LDS
// cycles and 10 of every 42 vector adds on operands; it measured 15.1 - 15.4 ms at C2 against 15.0 - 15.2 ms for this one on
// the same box, profiles/r03_jsd_lut.txt).  The scalar loads share lgkmcnt with the lookups and come back out of order: the next
// word's row counts are requested before the current word's lookups and waited for behind them with lgkmcnt(0) (the
// compiler does not know about them; its own counted waits only ever wait longer because of them).
typedef uint32_t po_u32x16 __attribute__((ext_vector_type(16)));
constexpr int kRowsMirrorCols = 64;                     // columns per transposition round of the mirrored tile
constexpr int kRowsMirrorBytes = kRowsMirrorCols * kMirrorLdsStride * 8;
constexpr int kRowsLdsBytes = kRowsMirrorBytes > kLutBytes ? kRowsMirrorBytes : kLutBytes;

template <int RW> struct rows_vec;
template <> struct rows_vec<16> {
    typedef po_u32x16 type;
    static __device__ __forceinline__ type load(const uint32_t* p) { type v; asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p)); return v; }
};

// RW = 16 rows per wave: 8 waves per tile (512 lanes), 4 waves per SIMD at two workgroups per CU.  (8 rows per wave, 16 waves per
// tile and 8 per SIMD measured 15.3 - 15.7 ms: more waves do not buy more overlap.)
template <typename OUT, int RW>
__global__ __launch_bounds__(64 * TM / RW, 4) void jsd_lut_rows_kernel(po_tile_args A, const uint32_t* __restrict__ ct,
                                                              const uint32_t* __restrict__ ctb, const double* __restrict__ lut,
                                                              const unsigned long long* __restrict__ cls,
                                                              const double2* __restrict__ clsk) {
    typedef typename rows_vec<RW>::type AV;
    constexpr uint32_t NT = 64 * TM / RW, NW = TM / RW;
    constexpr int kPF = 4;                                                       // words of column counts in flight per lane
    extern __shared__ __align__(16) unsigned char smem[];
    double* tab = reinterpret_cast<double*>(smem);                               // [256][32]; the mirror scratch afterwards
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);     // (a per-launch table of the coordinates instead measured 0.3 % - not worth a launch)
    const unsigned long long ntot = cls[ti];
    if (ntot == 0 || cls[tj] != ntot) return;                                    // valu_tile_kernel<JSD> owns this tile
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;
    {
        const uint4* src = reinterpret_cast<const uint4*>(lut);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t v = t; v < kLutBytes / 16; v += NT) dst[v] = src[v];
    }
    double acc[RW][2];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r][0] = acc[r][1] = 0.0;

    const uint32_t* pa = ct + i0 + RW * wv;                                      // + k * npad: wave uniform
    const uint32_t* pb = ctb + j0;                                               // + k * npad, + 2 lane
    const uint32_t lb = 2 * lane;
    const uint32_t tbase = po_lds_addr(tab);
    // The word loop runs over the zero-padded width D8 (prep_counts_kernel pads Ct / Ctb to whole groups of 8 words; a padded
    // word looks up T[0 + 0] = 0), NOT over A.dim: a round takes four words, and clamping the operand pointers at the last REAL
    // word of a width that is not a multiple of 4 would add that word's lookup again for every surplus slot.
    const uint32_t dim8 = (A.dim + 7u) & ~7u;
    const uint32_t kmax = dim8 - 1;
    // (operand pointers advance by one word row per step, clamped at the last word: scalar adds, no 64-bit multiplies and no
    // vector address arithmetic in the loop; the table's LDS base goes onto the two column terms, not onto the 16 row terms -
    // the scalar unit issued half as many instructions as the vector unit before: SQ_INSTS_SALU 3.2e9 against 6.0e9)
    uint2 bq[kPF];
#pragma unroll
    for (int q = 0; q < kPF; ++q) bq[q] = *reinterpret_cast<const uint2*>(pb + (uint64_t)min((uint32_t)q, kmax) * A.npad + lb);
    const uint32_t* pb_pf = pb + (uint64_t)min((uint32_t)kPF, kmax) * A.npad;    // the word the next refill of the ring reads
    const uint32_t* pa_nx = pa + (uint64_t)min(2u, kmax) * A.npad;              // the word whose row counts are requested next
    AV a_c0 = rows_vec<RW>::load(pa);                                            // words are taken in pairs: one wait for the scalar
    AV a_c1 = rows_vec<RW>::load(pa + (uint64_t)min(1u, kmax) * A.npad);         // loads (a full drain of lgkmcnt) per 64 lookups
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a_c0), "+s"(a_c1) :: "memory");
    __syncthreads();                                                             // the table is in LDS

    // Groups of four lookups (two rows), two groups in flight.  The instruction stream of a group is written out: four address
    // adds, four lookups, ONE counted wait (everything but the four lookups just issued is back), the four float64 adds of
    // the previous group.  Left to the compiler the same work carried a wait in front of almost every add (23 - 32 per word);
    // every instruction, a wait included, takes an issue slot of its in-order wave, and this loop is bound by issue slots.
    // (The scalar loads of the next words' row counts are also counted by lgkmcnt and return out of order: they can only make
    // a counted wait wait longer, never shorter.)
    // One asm statement per pair of words (64 lookups per lane): nothing of the compiler's sits between the groups.
    // operand names: a<r> / e<r> row counts of the two words (scalar), b0 b1 / d0 d1 their column terms, c<2r+e> the accumulators,
    // n<i> / m<i> the two groups of lookup results in flight, t<i> address scratch
#define PO_G_ISSUE(W, T, RA, RB, B0, B1)                                                                  \
    "v_add_u32 %[t0], %[" #W #RA "], %[" #B0 "]\n\tv_add_u32 %[t1], %[" #W #RA "], %[" #B1 "]\n\t"          \
    "v_add_u32 %[t2], %[" #W #RB "], %[" #B0 "]\n\tv_add_u32 %[t3], %[" #W #RB "], %[" #B1 "]\n\t"          \
    "ds_read_b64 %[" #T "0], %[t0]\n\tds_read_b64 %[" #T "1], %[t1]\n\tds_read_b64 %[" #T "2], %[t2]\n\tds_read_b64 %[" #T "3], %[t3]\n\t"
#define PO_G_ADD(T, C0, C1, C2, C3)                                                                        \
    "v_add_f64 %[c" #C0 "], %[c" #C0 "], %[" #T "0]\n\tv_add_f64 %[c" #C1 "], %[c" #C1 "], %[" #T "1]\n\t"    \
    "v_add_f64 %[c" #C2 "], %[c" #C2 "], %[" #T "2]\n\tv_add_f64 %[c" #C3 "], %[c" #C3 "], %[" #T "3]\n\t"
#define PO_G_WAIT4 "s_waitcnt lgkmcnt(4)\n\t"
    // eight groups of one word; PREV: the accumulators of the group in flight when the word starts (PO_G_NONE for the first word)
#define PO_G_NONE
#define PO_G_WORD(W, B0, B1, PREV)                                                                         \
    PO_G_ISSUE(W, n, 0, 1, B0, B1) PREV                                                                    \
    PO_G_ISSUE(W, m, 2, 3, B0, B1) PO_G_WAIT4 PO_G_ADD(n, 0, 1, 2, 3)                                        \
    PO_G_ISSUE(W, n, 4, 5, B0, B1) PO_G_WAIT4 PO_G_ADD(m, 4, 5, 6, 7)                                        \
    PO_G_ISSUE(W, m, 6, 7, B0, B1) PO_G_WAIT4 PO_G_ADD(n, 8, 9, 10, 11)                                      \
    PO_G_ISSUE(W, n, 8, 9, B0, B1) PO_G_WAIT4 PO_G_ADD(m, 12, 13, 14, 15)                                    \
    PO_G_ISSUE(W, m, 10, 11, B0, B1) PO_G_WAIT4 PO_G_ADD(n, 16, 17, 18, 19)                                  \
    PO_G_ISSUE(W, n, 12, 13, B0, B1) PO_G_WAIT4 PO_G_ADD(m, 20, 21, 22, 23)                                  \
    PO_G_ISSUE(W, m, 14, 15, B0, B1) PO_G_WAIT4 PO_G_ADD(n, 24, 25, 26, 27)
#define PO_G_ROWS(W, V) [W##0] "s"(V[0]), [W##1] "s"(V[1]), [W##2] "s"(V[2]), [W##3] "s"(V[3]), [W##4] "s"(V[4]), [W##5] "s"(V[5]),     \
    [W##6] "s"(V[6]), [W##7] "s"(V[7]), [W##8] "s"(V[8]), [W##9] "s"(V[9]), [W##10] "s"(V[10]), [W##11] "s"(V[11]),                  \
    [W##12] "s"(V[12]), [W##13] "s"(V[13]), [W##14] "s"(V[14]), [W##15] "s"(V[15])
#define PO_G_ACC(R) [c##R] "+v"(accf[R])
    static_assert(RW == 16 && kPF == 4, "the group chains below are written for 16 rows per wave and rounds of four words");
    auto doubling = [&](uint32_t k) {                                            // folded operands (po_fold.hip); dbl_at is a multiple of 8
        if (k == A.dbl_at) {
#pragma unroll
            for (int r = 0; r < RW; ++r) { acc[r][0] *= 2.0; acc[r][1] *= 2.0; }
        }
    };
    // two words (k, k + 1) per step; `clamp` (the last two rounds only) keeps the operand pointers on the last word row
    auto pair_of_words = [&](uint32_t k, int q, auto clamp_tag) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        doubling(k);
        AV a_n0 = rows_vec<RW>::load(pa_nx);
        if (!CLAMP || k + 3 <= kmax) pa_nx += A.npad;
        AV a_n1 = rows_vec<RW>::load(pa_nx);
        if (!CLAMP || k + 4 <= kmax) pa_nx += A.npad;
        const uint32_t b0 = bq[q].x + tbase, b1 = bq[q].y + tbase, c0 = bq[q + 1].x + tbase, c1 = bq[q + 1].y + tbase;
        bq[q] = *reinterpret_cast<const uint2*>(pb_pf + lb);
        if (!CLAMP || k + kPF + 1 <= kmax) pb_pf += A.npad;
        bq[q + 1] = *reinterpret_cast<const uint2*>(pb_pf + lb);
        if (!CLAMP || k + kPF + 2 <= kmax) pb_pf += A.npad;
        double n0, n1, n2, n3, m0, m1, m2, m3;
        uint32_t t0, t1, t2, t3;
        double* accf = &acc[0][0];                        // c<2r+e> = acc[r][e]
        // word k, then word k + 1 without a drain (dbl_at is even: no doubling can fall between the two): the first group of the
        // second word is issued before the last group of the first one is accumulated; the very last group is accumulated behind
        // the wait for the scalar loads of the next pair (lgkmcnt(0): everything is back)
        asm volatile(PO_G_WORD(a, b0, b1, PO_G_NONE)
                     PO_G_WORD(e, d0, d1, PO_G_WAIT4 PO_G_ADD(m, 28, 29, 30, 31))
                     "s_waitcnt lgkmcnt(0)\n\t" PO_G_ADD(m, 28, 29, 30, 31)
                     : PO_G_ACC(0), PO_G_ACC(1), PO_G_ACC(2), PO_G_ACC(3), PO_G_ACC(4), PO_G_ACC(5), PO_G_ACC(6), PO_G_ACC(7),
                       PO_G_ACC(8), PO_G_ACC(9), PO_G_ACC(10), PO_G_ACC(11), PO_G_ACC(12), PO_G_ACC(13), PO_G_ACC(14), PO_G_ACC(15),
                       PO_G_ACC(16), PO_G_ACC(17), PO_G_ACC(18), PO_G_ACC(19), PO_G_ACC(20), PO_G_ACC(21), PO_G_ACC(22), PO_G_ACC(23),
                       PO_G_ACC(24), PO_G_ACC(25), PO_G_ACC(26), PO_G_ACC(27), PO_G_ACC(28), PO_G_ACC(29), PO_G_ACC(30), PO_G_ACC(31),
                       [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3),
                       [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), "+s"(a_n0), "+s"(a_n1)
                     : PO_G_ROWS(a, a_c0), PO_G_ROWS(e, a_c1), [b0] "v"(b0), [b1] "v"(b1), [d0] "v"(c0), [d1] "v"(c1)
                     : "memory");
        a_c0 = a_n0;
        a_c1 = a_n1;
    };
    auto round = [&](uint32_t k0, auto clamp_tag) {
        pair_of_words(k0, 0, clamp_tag);
        pair_of_words(k0 + 2, 2, clamp_tag);
    };
#undef PO_G_ISSUE
#undef PO_G_ADD
#undef PO_G_WAIT4
#undef PO_G_NONE
#undef PO_G_WORD
#undef PO_G_ROWS
#undef PO_G_ACC
    uint32_t k0 = 0;
    for (; k0 + 2 * kPF < dim8; k0 += kPF) round(k0, std::false_type{});        // every pointer step stays inside the matrix
    for (; k0 < dim8; k0 += kPF) round(k0, std::true_type{});
    if (A.dbl_at != PO_NO_DOUBLING && A.dbl_at >= dim8) {                        // (folded widths are multiples of 8: dim8 == dim)
#pragma unroll
        for (int r = 0; r < RW; ++r) { acc[r][0] *= 2.0; acc[r][1] *= 2.0; }
    }

    // ---- epilogue: JSD = 1/2 (E_i + E_j - S) + ln 2,  S = Tsum/n - 2 ln n ----
    // Like the loop this part is paid in issue slots (its stores are asynchronous): the two constants of the class come from
    // classify_kernel, and tiles in the interior of the block - all but the last tile row / column - take a path without
    // per-lane bounds tests (uniform 64-bit row bases + 32-bit lane offsets, the diagonal fix on diagonal tiles only).
    const double2 kk = clsk[ti];
    const double inv_n = kk.x, two_ln_n = kk.y;
    const double* st0 = A.rowstat;
    const uint64_t jc = j0 + 2 * lane;
    const uint64_t ib = i0 + RW * wv;                                            // first row of the wave's block
    const double ej0 = st0[min(jc, A.npad - 1)], ej1 = st0[min(jc + 1, A.npad - 1)];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const double ei = st0[ib + r];
        acc[r][0] = fmax(0.5 * (ei + ej0 - fma(acc[r][0], inv_n, -two_ln_n)) + LN2, 0.0);
        acc[r][1] = fmax(0.5 * (ei + ej1 - fma(acc[r][1], inv_n, -two_ln_n)) + LN2, 0.0);
    }
    // (values at cancellation level - near-identical records - are noted for po_jsd_exact.hip: the smallest high word of the lane's
    //  values, one integer minimum per pair; the diagonal itself is left out)
    uint32_t lowhi = 0x7FF00000u;
    if (ti == tj) {                                                              // metric(x, x) / squareform diagonal
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (ib + r == jc) acc[r][0] = 0.0; else lowhi = min(lowhi, (uint32_t)__double2hiint(acc[r][0]));
            if (ib + r == jc + 1) acc[r][1] = 0.0; else lowhi = min(lowhi, (uint32_t)__double2hiint(acc[r][1]));
        }
    } else {
#pragma unroll
        for (int r = 0; r < RW; ++r)
            lowhi = min(lowhi, min((uint32_t)__double2hiint(acc[r][0]), (uint32_t)__double2hiint(acc[r][1])));
    }
    po_fix_note(A.fix, po_fix_hits(lowhi), ti, tj, RW * wv, RW);
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    double* lds = reinterpret_cast<double*>(smem);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const bool mirrors = po_tile_mirrors(A, ti, tj);                             // uniform over the workgroup
    const bool interior = i0 >= A.row_begin && i0 + TM <= n_rows && j0 >= A.col_begin && j0 + TN <= n_cols &&
                          ((A.ld_out | (j0 - A.col_begin)) & 1) == 0 && (reinterpret_cast<uintptr_t>(A.out) & (2 * sizeof(OUT) - 1)) == 0 &&
                          (!mirrors || (((A.ld_mirror | (i0 - A.row_begin)) & 1) == 0 &&
                                        (reinterpret_cast<uintptr_t>(A.mirror) & (2 * sizeof(OUT) - 1)) == 0));
    if (interior) {
        {   // the tile itself: a row of it is 1 KiB (float64) of one wave-instruction
            char* rowp = reinterpret_cast<char*>(out + (ib - A.row_begin) * A.ld_out + (j0 - A.col_begin));   // uniform
            const uint32_t loff = 2 * lane * (uint32_t)sizeof(OUT);
            const uint64_t pitch = A.ld_out * sizeof(OUT);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                po_store2(reinterpret_cast<OUT*>(rowp + loff), acc[r][0], acc[r][1]);
                rowp += pitch;
            }
        }
        if (!mirrors) return;
        // the transposed tile, 64 columns at a time through LDS (the table is dead): full 1 KiB row pieces again
        const uint32_t loff = 2 * lane * (uint32_t)sizeof(OUT);
        const uint64_t mpitch = A.ld_mirror * sizeof(OUT);
#pragma unroll
        for (int q = 0; q < TN / kRowsMirrorCols; ++q) {
            __syncthreads();                               // the table (q = 0) / the previous round has been read by everybody
            if ((lane >> 5) == (uint32_t)q) {              // the lanes holding columns [64 q, 64 q + 64)
                double* w = lds + 2 * (lane & 31) * kMirrorLdsStride + RW * wv;
#pragma unroll
                for (int r = 0; r < RW; r += 2) {
                    *reinterpret_cast<double2*>(w + r) = make_double2(acc[r][0], acc[r + 1][0]);
                    *reinterpret_cast<double2*>(w + kMirrorLdsStride + r) = make_double2(acc[r][1], acc[r + 1][1]);
                }
            }
            __syncthreads();
            char* mrow = reinterpret_cast<char*>(mir + (j0 + kRowsMirrorCols * q + wv - A.col_begin) * A.ld_mirror + (i0 - A.row_begin));   // uniform
            const double* rd = lds + wv * kMirrorLdsStride + 2 * lane;
#pragma unroll
            for (int jq = 0; jq < kRowsMirrorCols / (int)NW; ++jq) {             // one wave per transposed row: wv, wv + 8, ...
                const double2 w = *reinterpret_cast<const double2*>(rd + jq * (int)NW * kMirrorLdsStride);
                po_store2(reinterpret_cast<OUT*>(mrow + loff), w.x, w.y);
                mrow += NW * mpitch;
            }
        }
        return;
    }
    // ---- tiles on the edge of the block: every bound tested ----
    const bool vec_out = (A.ld_out & 1) == 0 && ((j0 - A.col_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.out) & (2 * sizeof(OUT) - 1)) == 0 && j0 >= A.col_begin;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const uint64_t i = ib + r;
        if (i < A.row_begin || i >= n_rows) continue;
        OUT* row = out + (i - A.row_begin) * A.ld_out;
        if (vec_out && jc + 1 < n_cols) {
            po_store2(row + (jc - A.col_begin), acc[r][0], acc[r][1]);
        } else {
            if (jc >= A.col_begin && jc < n_cols) po_out_store(&row[jc - A.col_begin], (OUT)acc[r][0]);
            if (jc + 1 >= A.col_begin && jc + 1 < n_cols) po_out_store(&row[jc + 1 - A.col_begin], (OUT)acc[r][1]);
        }
    }
    if (!mirrors) return;
    const bool vec_mir = (A.ld_mirror & 1) == 0 && ((i0 - A.row_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.mirror) & (2 * sizeof(OUT) - 1)) == 0 && i0 >= A.row_begin;
    for (int q = 0; q < TN / kRowsMirrorCols; ++q) {
        __syncthreads();
        if ((lane >> 5) == (uint32_t)q) {
            const uint32_t c = 2 * (lane & 31);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                lds[c * kMirrorLdsStride + RW * wv + r] = acc[r][0];
                lds[(c + 1) * kMirrorLdsStride + RW * wv + r] = acc[r][1];
            }
        }
        __syncthreads();
        for (uint32_t jq = wv; jq < kRowsMirrorCols; jq += NW) {
            const uint64_t j = j0 + kRowsMirrorCols * q + jq;
            if (j < A.col_begin || j >= n_cols) continue;
            const double2 w = *reinterpret_cast<const double2*>(lds + jq * kMirrorLdsStride + 2 * lane);
            const uint64_t i = i0 + 2 * lane;
            OUT* row = mir + (j - A.col_begin) * A.ld_mirror;
            if (vec_mir && i + 1 < n_rows) {
                po_store2(row + (i - A.row_begin), w.x, w.y);
            } else {
                if (i >= A.row_begin && i < n_rows) po_out_store(&row[i - A.row_begin], (OUT)w.x);
                if (i + 1 >= A.row_begin && i + 1 < n_rows) po_out_store(&row[i + 1 - A.row_begin], (OUT)w.y);
            }
        }
    }
}

}  // namespace

size_t po_jsd_lut_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return po_round_up(dim, 8) * npad * sizeof(uint32_t)      // Ct
           + po_round_up(npad / 128 * sizeof(unsigned long long), 16)   // cls
           + kLutBytes + 256                                    // replicated T + maxcount
           + po_round_up(dim, 8) * npad * sizeof(uint32_t)      // Ct with the lanes' table copies baked in (rows kernel)
           + npad / 128 * sizeof(double2);                      // clsk
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
    uint32_t* ctb = reinterpret_cast<uint32_t*>(base + kLutBytes + 256);
    double2* clsk = reinterpret_cast<double2*>(base + kLutBytes + 256 + po_round_up(dim, 8) * npad * sizeof(uint32_t));
    PO_HIP(hipMemsetAsync(maxcount, 0, sizeof(uint32_t), ctx->stream));
    dim3 grid((uint32_t)(npad / 64), (dim + 63) / 64);
    hipLaunchKernelGGL(prep_counts_kernel<false>, grid, dim3(256), 0, ctx->stream, d_counts, n, dim, npad, ct, ctb, maxcount);
    hipLaunchKernelGGL(prep_counts_kernel<true>, grid, dim3(256), 0, ctx->stream, d_counts, n, dim, npad, ct, ctb, maxcount);
    PO_CHECK_LAUNCH("prep_counts_kernel");
    hipLaunchKernelGGL(lut_table_kernel, dim3(kLutEntries), dim3(32), 0, ctx->stream, lut, maxcount);
    PO_CHECK_LAUNCH("lut_table_kernel");
    hipLaunchKernelGGL(classify_kernel, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, maxcount, d_wsum, cls, clsk);
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
    const uint32_t* ctb = reinterpret_cast<const uint32_t*>(base + kLutBytes + 256);
    const double2* clsk = reinterpret_cast<const double2*>(base + kLutBytes + 256 + po_round_up(a.dim, 8) * a.npad * sizeof(uint32_t));
    (void)n;
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    constexpr int RWL = 16;
    const size_t shmem = kRowsLdsBytes;
    if (a.out_f32) {
        PO_SHMEM(ctx, (jsd_lut_rows_kernel<float, RWL>), shmem);
        hipLaunchKernelGGL((jsd_lut_rows_kernel<float, RWL>), dim3((uint32_t)nblocks), dim3(64 * TM / RWL), shmem, ctx->stream, a, ct, ctb, lut, cls, clsk);
    } else {
        PO_SHMEM(ctx, (jsd_lut_rows_kernel<double, RWL>), shmem);
        hipLaunchKernelGGL((jsd_lut_rows_kernel<double, RWL>), dim3((uint32_t)nblocks), dim3(64 * TM / RWL), shmem, ctx->stream, a, ct, ctb, lut, cls, clsk);
    }
    PO_CHECK_LAUNCH("jsd_lut_rows_kernel");
    return PO_OK;
}
