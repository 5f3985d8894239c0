// Stage 2, Kendall tau on the int8 matrix cores (word spaces of at most 256 words, i.e. k <= 4).
//
// Same quantity as kt_tile_kernel (phylodist.KT, /root/reference/phylopackage/core/phylodist.py:71-74; see
// po_kt.hip for the tie algebra): per record pair only
//     S = sum_{p<q} sgn(x_p - x_q) sgn(y_p - y_q) = < sigma(x), sigma(y) >
// is needed, the dot product of the two records' PAIR-SIGN vectors sigma(x)_{(p,q)} = sgn(x_q - x_p) in
// {-1,0,+1}^{D(D-1)/2}.  That is a Gram matrix over int8 data: exactly what v_mfma_i32_32x32x32_i8 computes.
// The sign vectors (32 640 bytes per record at D = 256) are never materialised in HBM: a 128 x 128 tile
// keeps the 256 records' rank rows (uint8, 64 KiB) in LDS; each lane owns one record and expands, per
// K step of 32 word pairs, 2 x 16 signs with packed 16-bit arithmetic (v_pk_sub_i16, clamp to [-1,1],
// v_perm_b32 repack), writes them to a double-buffered LDS sign tile, and the four waves feed them to the
// matrix cores.  The workgroup is wave-specialised: waves 0-7 (two lanes per record) are producers that
// expand round r+1 while waves 8-15, which share their SIMDs, are consumers feeding round r to the MFMAs
// (64 x 32 outputs each; the mirrored output tile is transposed through LDS in the epilogue).  Word pairs are enumerated as
// items (p, block of 16 consecutive q): 2 160 items for D = 256, 6 % padding; blocks lying wholly above
// their p need no masking (a flag bit in the item says so).  Measured: 1.1e10 pairs/s at N = 50 000, D = 256
// (112 ms), against 1.3e8 pairs/s for the O(D^2)-per-pair VALU kernel of po_kt.hip.
//
// Strand-symmetric records (count[w] == count[rc(w)], checked by po_fold.hip): sigma only needs one word per
// reverse-complement orbit, a word pair (A, B) of orbits counting m_A m_B times (m = 2 for a two-word orbit).
// The kept words are laid out [self-paired | representatives], the items are sorted by weight class 4, 2, 1,
// and the consumer waves double their accumulators where a class ends: S = 4 S4 + 2 S2 + S1 - exact integers,
// 704 items instead of 2 160 at D = 256 (49 ms).
#include "po_tiles.h"

#include <stdlib.h>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v2s __attribute__((ext_vector_type(2)));

constexpr int TM = 128, TN = 128;
constexpr int kThreads = 1024;              // 16 waves: PWAVES of them expand signs, the others run the MFMAs
// Two splits of the workgroup.  PWAVES = 8: two lanes per record expand 8 items per round (4 K-steps of 32 word
// pairs), 8 consumer waves of 64 x 32 outputs.  PWAVES = 12 (when the rank rows are short enough for the larger
// sign tiles to fit LDS, e.g. folded k = 4): three lanes per record expand 12 items per round (6 K-steps), 4
// consumer waves of 64 x 64 outputs - the kernel is paced by the sign expansion, so more of the lanes go there.
// KSV = K-steps (32 word pairs = 2 items each) per round, i.e. per workgroup barrier: 4 (sign tiles of 2 x 36 KiB) or,
// when the rank rows leave room in LDS, 6 (2 x 52 KiB) - fewer barriers per tile.
__host__ __device__ constexpr int kt_sig_stride(int ksv) { return ksv * 32 + 16; }   // +16: conflict-free b128 rows

// rank8[r][c] = (uint8) lessrank[r][src ? src[c] : c], rows of `row_bytes` columns (zero beyond the words / records).
// src = the folded column order of po_fold.hip: ranks among all D words keep the order and ties of the kept words.
__global__ __launch_bounds__(256) void rank8_kernel(const uint32_t* __restrict__ lessrank, uint64_t n, uint32_t dim,
                                                    uint64_t npad, const uint32_t* __restrict__ src, uint32_t row_bytes,
                                                    uint8_t* __restrict__ rank8) {
    const uint64_t total = npad * row_bytes;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / row_bytes;
        const uint32_t c = (uint32_t)(i - r * row_bytes);
        const uint32_t w = src ? src[c] : c;                           // 0xFFFFFFFF: padding column
        rank8[i] = (r < n && w < dim) ? (uint8_t)lessrank[r * dim + w] : (uint8_t)0;
    }
}

// 4 packed bytes of ranks -> 4 packed sign bytes of (x_q - x_p).  one2 / mone2 = {1,1} / {-1,-1} in VGPRs
// (packed 16-bit immediates are not inline constants); the clamps are pinned to v_pk_min/max_i16 because
// hipcc otherwise scalarises them into compare + select pairs.
__device__ __forceinline__ uint32_t sign4(uint32_t w, uint32_t xp2, uint32_t one2, uint32_t mone2) {
    uint32_t lo = w & 0x00FF00FFu, hi = (w >> 8) & 0x00FF00FFu;
    asm("v_pk_sub_i16 %0, %0, %1\n\tv_pk_min_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %3" : "+v"(lo) : "v"(xp2), "v"(one2), "v"(mone2));
    asm("v_pk_sub_i16 %0, %0, %1\n\tv_pk_min_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %3" : "+v"(hi) : "v"(xp2), "v"(one2), "v"(mone2));
    // bytes: out0 = lo.b0, out1 = hi.b0, out2 = lo.b2, out3 = hi.b2   (perm: src1 = bytes 0-3, src0 = bytes 4-7)
    return __builtin_amdgcn_perm(hi, lo, 0x06020400u);
}

template <typename OUT, int PWAVES, int KSV>
__global__ __launch_bounds__(kThreads, 4) void kt_mfma_tile_kernel(po_tile_args A, const uint8_t* __restrict__ rank8,
                                                                   const uint16_t* __restrict__ items, uint32_t n_items,
                                                                   uint32_t row_bytes, uint32_t dim_full, uint32_t dbl1,
                                                                   uint32_t dbl2) {
    // A.dim = number of words the items range over (the folded count when dbl1/dbl2 are set), row_bytes = length of a
    // rank row in memory (a multiple of 16 or A.dim), dim_full = D of the records (tie algebra of the epilogue).
    constexpr int kSigStride = kt_sig_stride(KSV);
    constexpr int KS = 2 * KSV / (PWAVES / 4);                         // items per producer lane and round (4 or 6)
    constexpr int NB = PWAVES == 8 ? 1 : 2;                            // 32-column blocks per consumer wave
    constexpr int CH = (PWAVES == 8 && KSV == 4) ? 4 : 3;              // K-steps whose fragments are in flight at once
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t rstride = row_bytes + 16;                           // rank row stride in LDS (bytes)
    unsigned char* ranks = smem;                                       // [256][rstride]
    unsigned char* sigma = smem + ((256 * rstride + 15) & ~15u);       // [2][256][kSigStride]
    uint16_t* litems = reinterpret_cast<uint16_t*>(sigma + 2 * 256 * kSigStride);   // [n_items] word-pair items

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    const bool producer = wave < PWAVES;                               // wave-uniform role
    const uint32_t cw_ = producer ? 0u : wave - PWAVES;                // consumer wave -> 64 rows x 32 NB columns of the tile
    const uint32_t wr = PWAVES == 8 ? cw_ >> 2 : cw_ >> 1, wc = PWAVES == 8 ? cw_ & 3 : cw_ & 1;
    const uint32_t half = __builtin_amdgcn_readfirstlane(t >> 8) & 3;  // producer wave: which part of a round's items (uniform)
    const uint32_t lr = lane & 31, lh = lane >> 5;

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;
    const bool mirror = po_tile_mirrors(A, ti, tj);

    // ---- producer lane = one record: rank row into LDS (rows 0..127 = tile rows, 128..255 = tile columns) ----
    if (producer && half == 0) {
        const uint64_t rec = (t < 128) ? i0 + t : j0 + (t - 128);      // < npad: padded rows are zero
        const uint8_t* src = rank8 + rec * row_bytes;
        unsigned char* dst = ranks + t * rstride;
        if ((row_bytes & 15u) == 0) {
            for (uint32_t d = 0; d < row_bytes; d += 16) *reinterpret_cast<uint4*>(dst + d) = *reinterpret_cast<const uint4*>(src + d);
        } else {
            for (uint32_t d = 0; d < row_bytes; ++d) dst[d] = src[d];
            for (uint32_t d = row_bytes; d < ((row_bytes + 15u) & ~15u); ++d) dst[d] = 0;
        }
    }
    for (uint32_t i = t; i < n_items; i += kThreads) litems[i] = items[i];
    __syncthreads();

    v16i g[2][NB];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[m][nb][e] = 0;

    // sign expansion of KS K-steps (2 items each) for this lane's record into sign buffer `buf`.
    // MASKED = false for the rounds whose 16-q blocks lie entirely above their p (no byte to suppress).
    const unsigned char* myrank = ranks + (t & 255) * rstride;
    uint32_t one2, mone2;
    asm volatile("v_mov_b32 %0, 0x00010001" : "=v"(one2));
    asm volatile("v_mov_b32 %0, -1" : "=v"(mone2));
    auto expand = [&](uint32_t item0, uint32_t buf) {
        unsigned char* dst = sigma + (buf * 256 + (t & 255)) * kSigStride;
        // this wave's KS item codes (wave uniform -> scalar registers): (p << 8) | partial flag (bit 7) | q block
        uint32_t cw[KS / 2];
        uint32_t flags = 0;
#pragma unroll
        for (int i = 0; i < KS / 2; ++i) {
            cw[i] = (uint32_t)__builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t*>(litems + item0 + half * KS)[i]);
            flags |= cw[i];
        }
        const bool masked = (flags & 0x00800080u) != 0u;               // blocks lying wholly above their p need no masking
        // all LDS reads of the lane's KS items first (the sign-tile stores below may alias them for the
        // compiler, which would otherwise serialise read -> compute -> store item by item)
        uint32_t pp[KS], qq[KS], xps[KS];
        uint4 ws[KS];
#pragma unroll
        for (int it4 = 0; it4 < KS; ++it4) {                           // this lane's half of the round's 2 KS items
            const uint32_t code = (cw[it4 >> 1] >> (16 * (it4 & 1))) & 0xFFFFu;   // wave uniform: (p << 8) | qblock
            pp[it4] = code >> 8;
            qq[it4] = code & 0x0Fu;
            xps[it4] = myrank[pp[it4]];
            ws[it4] = *reinterpret_cast<const uint4*>(myrank + qq[it4] * 16);
        }
#pragma unroll
        for (int it4 = 0; it4 < KS; ++it4) {
            const uint32_t it = half * KS + it4;
            const uint32_t p = pp[it4], qb = qq[it4];
            const uint32_t xp2 = xps[it4] | (xps[it4] << 16);
            const uint4 w = ws[it4];
            uint32_t sg[4] = {sign4(w.x, xp2, one2, mone2), sign4(w.y, xp2, one2, mone2),
                              sign4(w.z, xp2, one2, mone2), sign4(w.w, xp2, one2, mone2)};
            if (masked) {     // only q in (p, dim) counts: bytes [first, last) of the block survive (uniform)
                const uint32_t q0 = qb * 16;
                const uint32_t first = (p + 1 > q0) ? min(p + 1 - q0, 16u) : 0u;
                const uint32_t last = (A.dim - q0 < 16u) ? A.dim - q0 : 16u;
#pragma unroll
                for (uint32_t wi = 0; wi < 4; ++wi) {
                    const uint32_t lo_b = first > 4 * wi ? min(first - 4 * wi, 4u) : 0u;    // leading bytes to clear
                    const uint32_t hi_b = last > 4 * wi ? min(last - 4 * wi, 4u) : 0u;      // bytes before `last` kept
                    const uint32_t keep_lo = lo_b >= 4 ? 0u : (0xFFFFFFFFu << (8 * lo_b));
                    const uint32_t keep_hi = hi_b >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi_b)) - 1u);
                    sg[wi] &= keep_lo & keep_hi;
                }
            }
            *reinterpret_cast<uint4*>(dst + it * 16) = make_uint4(sg[0], sg[1], sg[2], sg[3]);
        }
    };

    const uint32_t n_rounds = n_items / (2 * KSV);                      // items are padded to whole rounds of 2 KSV
    auto consume = [&](uint32_t buf) {
        const unsigned char* sa = sigma + (buf * 256 + wr * 64 + lr) * kSigStride + 16 * lh;
        const unsigned char* sb = sigma + (buf * 256 + 128 + wc * 32 * NB + lr) * kSigStride + 16 * lh;
#pragma unroll
        for (int k0 = 0; k0 < KSV; k0 += CH) {
            v4i a[CH][2], b[CH][NB];                                   // the fragment reads of CH K-steps in flight at once
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[ks][m] = *reinterpret_cast<const v4i*>(sa + m * 32 * kSigStride + (k0 + ks) * 32);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[ks][nb] = *reinterpret_cast<const v4i*>(sb + nb * 32 * kSigStride + (k0 + ks) * 32);
            }
#pragma unroll
            for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        g[m][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks][m], b[ks][nb], g[m][nb], 0, 0, 0);
        }
    };
    // rounds [0, n_full_rounds) hold only whole blocks (no masking code in the hot loop), the rest may be partial.
    // Producer waves write round r+1 into one sign buffer while the consumer waves, which share their SIMDs,
    // feed round r from the other buffer to the matrix cores: VALU expansion and MFMA overlap.
    // Folded operands (po_fold.hip): the items come in three classes - both words stand for two-word orbits
    // (weight 4), one does (2), none does (1) - in that order; doubling the accumulators where a class ends
    // gives 4 S4 + 2 S2 + S1 without touching the sign expansion.
    auto double_sums = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) g[m][nb][e] <<= 1;
    };
    if (producer) expand(0, 0);
    __syncthreads();
    for (uint32_t r = 0; r < n_rounds; ++r) {
        const uint32_t buf = r & 1;
        if (producer) {
            if (r + 1 < n_rounds) expand((r + 1) * 2 * KSV, buf ^ 1);
        } else {
            if (r == dbl1) double_sums();
            if (r == dbl2) double_sums();
            consume(buf);
        }
        __syncthreads();
    }
    if (producer) return;
    if (dbl1 != PO_NO_DOUBLING && dbl1 >= n_rounds) double_sums();
    if (dbl2 != PO_NO_DOUBLING && dbl2 >= n_rounds) double_sums();

    // ---- epilogue: tau = S / sqrt((T - t_r)(T - t_c)), KT = 1 - (1 - tau), 0 when a factor vanishes -----
    // The mirrored tile is transposed through wave-private LDS (the rank rows and sign tiles are no longer
    // needed): a second MFMA with swapped operands would be exact too, but it measured 17 % of the kernel.
    const double T = 0.5 * (double)dim_full * ((double)dim_full - 1.0);
    const double* ties = A.rowstat + 3 * A.npad;
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const uint64_t ri = i0 + wr * 64, cj = j0 + wc * 32 * NB;
    double* wl = reinterpret_cast<double*>(smem) + cw_ * (32 * 33);            // one 8.4 KiB scratch per consumer wave
    double drs[2][16];                                     // every load before the first store (shared in-order vmcnt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) drs[m][reg] = T - ties[min(ri + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh, A.npad - 1)];
    double dcs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dcs[nb] = T - ties[min(cj + nb * 32 + lr, A.npad - 1)];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const uint64_t c = cj + nb * 32 + lr;
        const double dc = dcs[nb];
        const bool c_ok = c >= A.col_begin && c < n_cols;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                const uint64_t rr = ri + m * 32 + rl;
                const double dr = drs[m][reg];
                double v;
                if (dr == 0.0 || dc == 0.0) {
                    v = 1.0 - 1.0;
                } else {
                    const double tau = (double)g[m][nb][reg] / sqrt(dr * dc);
                    v = 1.0 - (1.0 - tau);
                }
                if (c_ok && rr >= A.row_begin && rr < n_rows) out[(rr - A.row_begin) * A.ld_out + (c - A.col_begin)] = (OUT)v;
                if (mirror) wl[lr * 33 + rl] = v;
            }
            if (mirror) {                                  // LDS operations of one wave execute in order
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const uint32_t jr = it * 2 + lh;
                    const double w = wl[jr * 33 + lr];
                    const uint64_t cm = cj + nb * 32 + jr, rr = ri + m * 32 + lr;
                    if (cm >= A.col_begin && cm < n_cols && rr >= A.row_begin && rr < n_rows)
                        mir[(cm - A.col_begin) * A.ld_mirror + (rr - A.row_begin)] = (OUT)w;
                }
            }
        }
    }
}

}  // namespace

bool po_kt_mfma_supported(uint32_t dim) { return dim <= 256; }
// folded layout [self-paired words | orbit representatives]: the 16-word q blocks must not straddle the two regions
bool po_kt_mfma_fold_supported(uint32_t dim, uint32_t n_selfs) { return dim <= 256 && (n_selfs % 16u) == 0; }

size_t po_kt_mfma_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    const size_t items = (size_t)dim * (dim / 16 + 2) + 64;
    return npad * po_round_up(dim, 16) + items * sizeof(uint16_t) + 512;
}

// ws layout: rank8[npad][row_bytes] | items.
// fold_src != NULL: the records are reverse-complement symmetric (checked by po_rc_fold); only the words
// fold_src[0 .. n_selfs + n_pairs) are kept, the first n_selfs standing for themselves, the others for two words each.
int po_launch_kt_mfma_prep(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, uint64_t npad, void* ws,
                           const uint32_t* fold_src, uint32_t n_selfs, uint32_t n_pairs, po_kt_mfma_plan* plan) {
    const uint32_t words = fold_src ? n_selfs + n_pairs : dim;        // words the items range over
    const uint32_t row_bytes = fold_src ? (uint32_t)po_round_up(words, 16) : dim;
    // Rounds of 4 K-steps and 8 expanding waves are the default.  Both alternatives were built and measured at
    // N = 50 000, folded k = 4 (45.2 ms): 6 K-steps per round (fewer barriers, needs the larger sign tiles to fit LDS)
    // 46.2 ms, 12 expanding waves 48.5 ms - neither the barrier count nor the number of expanding lanes paces
    // the kernel.  The variants stay in the source, not selectable at run time.
    const bool want12 = false, want6 = false;
    const bool fits6 = 256 * (size_t)(row_bytes + 16) + 2 * 256 * (size_t)kt_sig_stride(6) + 4096 <= 160 * 1024;
    const int ksv = (fits6 && want6) ? 6 : 4;
    const int pwaves = (want12 && ksv == 6) ? 12 : 8;
    const uint32_t round_items = 2 * (uint32_t)ksv;
    uint8_t* rank8 = static_cast<uint8_t*>(ws);
    uint16_t* d_items = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(ws) + ((npad * row_bytes + 255) & ~(uint64_t)255));
    hipLaunchKernelGGL(rank8_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_lessrank, n, dim, npad, fold_src, row_bytes, rank8);
    PO_CHECK_LAUNCH("rank8_kernel");
    // word pairs as items (p, block of 16 q): every block that contains a q in (p, words).  Per weight class
    // (4, 2, 1; a single class when not folded): blocks lying wholly inside (p, words) first, then the partial
    // ones (flag bit 7: first block of a p, last block of the row), then fully masked padding up to a whole round.
    static thread_local uint16_t host_items[256 * 18 + 96];
    uint32_t cnt = 0;
    const uint32_t nblk = (words + 15) / 16;
    uint32_t class_start[4] = {0, 0, 0, 0};
    const int n_classes = fold_src ? 3 : 1;
    for (int cls = 0; cls < n_classes; ++cls) {
        const uint32_t want = fold_src ? (4u >> cls) : 0u;             // 4, 2, 1
        class_start[cls] = cnt / round_items;
        for (int partial = 0; partial < 2; ++partial)
            for (uint32_t p = 0; p + 1 < words; ++p)
                for (uint32_t qb = (p + 1) / 16; qb < nblk; ++qb) {
                    if (fold_src) {
                        const uint32_t w = (p < n_selfs ? 1u : 2u) * (qb * 16 < n_selfs ? 1u : 2u);
                        if (w != want) continue;
                    }
                    const bool whole = qb * 16 > p && qb * 16 + 16 <= words;
                    if (whole == (partial == 0)) host_items[cnt++] = (uint16_t)((p << 8) | (whole ? 0u : 0x80u) | qb);
                }
        while (cnt % round_items) host_items[cnt++] = (uint16_t)(((words - 1) << 8) | 0x80u);   // fully masked padding
    }
    while (cnt == 0) for (uint32_t i = 0; i < round_items; ++i) host_items[cnt++] = (uint16_t)(((words - 1) << 8) | 0x80u);
    class_start[n_classes] = cnt / round_items;
    PO_HIP(hipMemcpyAsync(d_items, host_items, cnt * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));                        // host_items is reused by the next call
    plan->n_items = cnt;
    plan->words = words;
    plan->row_bytes = row_bytes;
    plan->pwaves = (uint32_t)pwaves;
    plan->ksteps = (uint32_t)ksv;
    plan->dbl1 = fold_src ? class_start[1] : PO_NO_DOUBLING;          // == number of rounds when the later classes are empty
    plan->dbl2 = fold_src ? class_start[2] : PO_NO_DOUBLING;
    return PO_OK;
}

int po_launch_kt_mfma_tiles(po_ctx* ctx, const po_tile_args& a_in, const void* ws, const po_kt_mfma_plan& plan, uint64_t* tiles) {
    po_tile_args a = a_in;
    const uint32_t dim_full = a.dim;
    a.dim = plan.words;
    const uint8_t* rank8 = static_cast<const uint8_t*>(ws);
    const uint16_t* d_items = reinterpret_cast<const uint16_t*>(static_cast<const uint8_t*>(ws) + ((a.npad * plan.row_bytes + 255) & ~(uint64_t)255));
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = ((256 * (plan.row_bytes + 16) + 15) & ~(size_t)15) + 2 * 256 * (size_t)kt_sig_stride((int)plan.ksteps) +
                         ((plan.n_items * 2 + 15) & ~(size_t)15);
    auto launch = [&](auto k) -> int {
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, rank8, d_items, plan.n_items,
                           plan.row_bytes, dim_full, plan.dbl1, plan.dbl2);
        return PO_OK;
    };
    int lrc;
    if (plan.pwaves == 12) lrc = a.out_f32 ? launch(kt_mfma_tile_kernel<float, 12, 6>) : launch(kt_mfma_tile_kernel<double, 12, 6>);
    else if (plan.ksteps == 6) lrc = a.out_f32 ? launch(kt_mfma_tile_kernel<float, 8, 6>) : launch(kt_mfma_tile_kernel<double, 8, 6>);
    else lrc = a.out_f32 ? launch(kt_mfma_tile_kernel<float, 8, 4>) : launch(kt_mfma_tile_kernel<double, 8, 4>);
    if (lrc) return lrc;
    PO_CHECK_LAUNCH("kt_mfma_tile_kernel");
    return PO_OK;
}
