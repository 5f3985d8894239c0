// Stage 2, Kendall tau on the int8 matrix cores for word spaces above 256 words (k = 5..7).
//
// Same quantity and the same scheme as po_kt_mfma.hip (phylodist.KT,
// /root/reference/phylopackage/core/phylodist.py:71-74): S = < sigma(x), sigma(y) > over the pair-sign vectors,
// expanded on the fly by producer waves and fed to v_mfma_i32_32x32x32_i8 by consumer waves.  What changes with D:
//   * ranks need 16 bits (uint16 rows, differences by v_pk_sub_i16 directly, no byte unpacking);
//   * the rank rows of a tile's 256 records no longer fit LDS, so the word space is cut into PANELS of 64 words and
//     the word pairs are walked panel pair by panel pair (P <= Q): two 256 x 64 rank panels live in LDS at a time.
//     An off-diagonal panel pair is 64 x 4 whole items (p, block of 16 q), one p per producer half and round; a
//     diagonal pair (p < q inside one panel) uses a small precomputed item list with masks, as po_kt_mfma.hip does.
//   * folded operands (po_fold.hip, layout [self-paired words | orbit representatives], the self-paired region a
//     whole number of panels): panel pairs are visited by weight class 4, 2, 1 and the accumulators doubled in between.
// Exact integers; bit-identical to kt_tile_kernel (po_kt.hip), which remains the reference and the fallback.
#include "po_tiles.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TM = 128, TN = 128;
constexpr int kThreads = 1024;              // waves 0-7 expand signs (two lanes per record), waves 8-15 run the MFMAs
constexpr int PW = 64;                      // words per panel
constexpr int KS = 4;                       // K steps (32 word pairs each) per barrier = 8 items of 16 word pairs
constexpr int kSigStride = KS * 32 + 16;    // bytes per record in the sign tile (+16: conflict-free b128 rows)
constexpr int kRankStride = PW * 2 + 16;    // bytes per record in a rank panel

// rank16[r][c] = (uint16) lessrank[r][src ? src[c] : c] for c < row_words (0 for padding columns / records)
__global__ __launch_bounds__(256) void rank16_kernel(const uint32_t* __restrict__ lessrank, uint64_t n, uint32_t dim,
                                                     uint64_t npad, const uint32_t* __restrict__ src, uint32_t src_len,
                                                     uint32_t row_words, uint16_t* __restrict__ rank16) {
    const uint64_t total = npad * row_words;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / row_words;
        const uint32_t c = (uint32_t)(i - r * row_words);
        const uint32_t w = src ? (c < src_len ? src[c] : 0xFFFFFFFFu) : c;
        rank16[i] = (r < n && w < dim) ? (uint16_t)lessrank[r * dim + w] : (uint16_t)0;
    }
}

// two dwords = four uint16 ranks -> four packed sign bytes of (x_q - x_p)
__device__ __forceinline__ uint32_t sign4_u16(uint32_t a, uint32_t b, uint32_t xp2, uint32_t one2, uint32_t mone2) {
    asm("v_pk_sub_i16 %0, %0, %1\n\tv_pk_min_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %3" : "+v"(a) : "v"(xp2), "v"(one2), "v"(mone2));
    asm("v_pk_sub_i16 %0, %0, %1\n\tv_pk_min_i16 %0, %0, %2\n\tv_pk_max_i16 %0, %0, %3" : "+v"(b) : "v"(xp2), "v"(one2), "v"(mone2));
    return __builtin_amdgcn_perm(b, a, 0x06040200u);     // bytes a.b0, a.b2, b.b0, b.b2
}

template <typename OUT>
__global__ __launch_bounds__(kThreads, 4) void kt_panel_tile_kernel(po_tile_args A, const uint16_t* __restrict__ rank16,
                                                                    uint32_t row_words, uint32_t words, uint32_t dim_full,
                                                                    uint32_t self_panels, int folded,
                                                                    const uint16_t* __restrict__ diag_items,
                                                                    uint32_t n_diag_items) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* rankP = smem;                                       // [256][kRankStride]
    unsigned char* rankQ = rankP + 256 * kRankStride;
    unsigned char* sigma = rankQ + 256 * kRankStride;                  // [2][256][kSigStride]
    uint16_t* litems = reinterpret_cast<uint16_t*>(sigma + 2 * 256 * kSigStride);

    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wave = t >> 6;
    const bool producer = wave < 8;                                    // wave-uniform role
    const uint32_t cw_ = wave & 7, wr = cw_ >> 2, wc = cw_ & 3;        // consumer wave -> 64 rows x 32 columns of the tile
    const uint32_t half = __builtin_amdgcn_readfirstlane(t >> 8) & 1;  // producer wave: which half of a round's items
    const uint32_t lr = lane & 31, lh = lane >> 5;
    const uint32_t rec_l = t & 255;                                    // producer lane's record: 0..127 rows, 128..255 columns

    uint32_t ti, tj;
    po_tile_coords(A, TM, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TM, j0 = (uint64_t)tj * TN;
    const bool mirror = po_tile_mirrors(A, ti, tj);
    const uint64_t my_rec = (rec_l < 128) ? i0 + rec_l : j0 + (rec_l - 128);   // < npad: padded rows are zero

    for (uint32_t i = t; i < n_diag_items; i += kThreads) litems[i] = diag_items[i];

    v16i g[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) g[m][e] = 0;

    uint32_t one2, mone2;
    asm volatile("v_mov_b32 %0, 0x00010001" : "=v"(one2));
    asm volatile("v_mov_b32 %0, -1" : "=v"(mone2));

    // one panel (64 words = 128 bytes per record) of this lane's record into LDS; producer half 0 only
    auto load_panel = [&](unsigned char* dst, uint32_t panel) {
        const uint4* src = reinterpret_cast<const uint4*>(rank16 + my_rec * row_words + panel * PW);
        uint4* d = reinterpret_cast<uint4*>(dst + rec_l * kRankStride);
#pragma unroll
        for (int v = 0; v < PW * 2 / 16; ++v) d[v] = src[v];
    };

    // sign expansion of this half's KS items of one round into sign buffer `buf`.
    //   off-diagonal panel pair: the half's four items are (p, q blocks 0..3) for p = 2 round + half
    //   diagonal panel pair: items from the list, (p << 8) | partial flag (bit 7) | q block
    auto expand = [&](const unsigned char* rp, const unsigned char* rq, bool diag, uint32_t round, uint32_t pw, uint32_t qw,
                      uint32_t buf) {
        unsigned char* dst = sigma + (buf * 256 + rec_l) * kSigStride;
        const unsigned char* myp = rp + rec_l * kRankStride;
        const unsigned char* myq = rq + rec_l * kRankStride;
        uint32_t pp[KS], qq[KS];
        bool masked = false;
        if (diag) {
            const uint2 c2 = *reinterpret_cast<const uint2*>(litems + round * 2 * KS + half * KS);
            const uint32_t cw[2] = {(uint32_t)__builtin_amdgcn_readfirstlane(c2.x), (uint32_t)__builtin_amdgcn_readfirstlane(c2.y)};
            masked = true;                                              // diagonal pairs are few: always the masking path
#pragma unroll
            for (int it4 = 0; it4 < KS; ++it4) {
                const uint32_t code = (cw[it4 >> 1] >> (16 * (it4 & 1))) & 0xFFFFu;
                pp[it4] = code >> 8;
                qq[it4] = code & 0x0Fu;
            }
        } else {
#pragma unroll
            for (int it4 = 0; it4 < KS; ++it4) { pp[it4] = round * 2 + half; qq[it4] = it4; }
            masked = pw < PW || qw < PW;                                // the last panel of the row may be short
        }
        // two items at a time: their LDS reads first (the sign-tile stores may alias them for the compiler, which
        // would otherwise serialise read -> compute -> store item by item), then the arithmetic
#pragma unroll
        for (int i2 = 0; i2 < KS; i2 += 2) {
            uint32_t xps[2];
            uint4 w0[2], w1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xps[j] = *reinterpret_cast<const uint16_t*>(myp + pp[i2 + j] * 2);
                w0[j] = *reinterpret_cast<const uint4*>(myq + qq[i2 + j] * 32);
                w1[j] = *reinterpret_cast<const uint4*>(myq + qq[i2 + j] * 32 + 16);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int it4 = i2 + j;
                const uint32_t it = half * KS + it4;
                const uint32_t xp2 = xps[j] | (xps[j] << 16);
                uint32_t sg[4] = {sign4_u16(w0[j].x, w0[j].y, xp2, one2, mone2), sign4_u16(w0[j].z, w0[j].w, xp2, one2, mone2),
                                  sign4_u16(w1[j].x, w1[j].y, xp2, one2, mone2), sign4_u16(w1[j].z, w1[j].w, xp2, one2, mone2)};
                if (masked) {     // bytes [first, last) of the block survive (uniform): q > p inside a diagonal pair, q and p inside the row
                    const uint32_t p = pp[it4], q0 = qq[it4] * 16;
                    uint32_t first = 0, last = qw > q0 ? min(qw - q0, 16u) : 0u;
                    if (diag) first = (p + 1 > q0) ? min(p + 1 - q0, 16u) : 0u;
                    if (p >= pw) last = 0;
#pragma unroll
                    for (uint32_t wi = 0; wi < 4; ++wi) {
                        const uint32_t lo_b = first > 4 * wi ? min(first - 4 * wi, 4u) : 0u;
                        const uint32_t hi_b = last > 4 * wi ? min(last - 4 * wi, 4u) : 0u;
                        const uint32_t keep_lo = lo_b >= 4 ? 0u : (0xFFFFFFFFu << (8 * lo_b));
                        const uint32_t keep_hi = hi_b >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi_b)) - 1u);
                        sg[wi] &= keep_lo & keep_hi;
                    }
                }
                *reinterpret_cast<uint4*>(dst + it * 16) = make_uint4(sg[0], sg[1], sg[2], sg[3]);
            }
        }
    };

    auto consume = [&](uint32_t buf) {
        const unsigned char* sa = sigma + (buf * 256 + wr * 64 + lr) * kSigStride + 16 * lh;
        const unsigned char* sb = sigma + (buf * 256 + 128 + wc * 32 + lr) * kSigStride + 16 * lh;
        v4i a[KS][2], b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < 2; ++m) a[ks][m] = *reinterpret_cast<const v4i*>(sa + m * 32 * kSigStride + ks * 32);
            b[ks] = *reinterpret_cast<const v4i*>(sb + ks * 32);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                g[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks][m], b[ks], g[m], 0, 0, 0);
            }
        }
    };
    auto double_sums = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[m][e] <<= 1;
    };

    const uint32_t n_panels = (words + PW - 1) / PW;
    const uint32_t n_diag_rounds = n_diag_items / (2 * KS);
    // weight classes of the folded layout: (representative, representative) = 4, (self, representative) = 2, (self, self) = 1
    const int n_classes = folded ? 3 : 1;
    for (int cls = 0; cls < n_classes; ++cls) {
        uint32_t p_lo = 0, p_hi = n_panels;
        if (folded) {
            if (cls == 0) p_lo = self_panels; else p_hi = self_panels;
            if (!producer && cls > 0) double_sums();
        }
        for (uint32_t pp = p_lo; pp < p_hi; ++pp) {
            uint32_t q_lo = pp, q_hi = n_panels;
            if (folded && cls == 1) q_lo = self_panels;
            if (folded && cls == 2) q_hi = self_panels;
            const uint32_t pw = min((uint32_t)PW, words - pp * PW);
            __syncthreads();                                           // everyone is done with the previous panels
            if (producer && half == 0) load_panel(rankP, pp);
            for (uint32_t qp = q_lo; qp < q_hi; ++qp) {
                const bool diag = qp == pp;
                const uint32_t qw = min((uint32_t)PW, words - qp * PW);
                const unsigned char* rq = diag ? rankP : rankQ;
                if (!diag) {
                    __syncthreads();                                   // the previous Q panel is no longer read
                    if (producer && half == 0) load_panel(rankQ, qp);
                }
                __syncthreads();                                       // panels are in LDS
                const uint32_t n_rounds = diag ? n_diag_rounds : PW / 2;
                if (producer) expand(rankP, rq, diag, 0, pw, qw, 0);
                __syncthreads();
                for (uint32_t r = 0; r < n_rounds; ++r) {
                    const uint32_t buf = r & 1;
                    if (producer) {
                        if (r + 1 < n_rounds) expand(rankP, rq, diag, r + 1, pw, qw, buf ^ 1);
                    } else {
                        consume(buf);
                    }
                    __syncthreads();
                }
            }
        }
    }
    if (producer) return;

    // ---- epilogue: tau = S / sqrt((T - t_r)(T - t_c)), KT = 1 - (1 - tau), 0 when a factor vanishes -----
    // the mirrored tile is transposed through wave-private LDS (rank panels and sign tiles are no longer needed:
    // the last round ended with a barrier)
    const double T = 0.5 * (double)dim_full * ((double)dim_full - 1.0);
    const double* ties = A.rowstat + 3 * A.npad;
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const uint64_t ri = i0 + wr * 64, cj = j0 + wc * 32;
    double* wl = reinterpret_cast<double*>(smem) + cw_ * (32 * 33);            // 8 consumer waves x 8.4 KiB
    const uint64_t c = cj + lr;
    const double dc = T - ties[min(c, A.npad - 1)];
    const double rsc = po_kt_rs(dc);
    double drs[2][16];                                     // every load before the first store (shared in-order vmcnt)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) drs[m][reg] = T - ties[min(ri + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh, A.npad - 1)];
    const bool c_ok = c >= A.col_begin && c < n_cols;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const uint64_t rr = ri + m * 32 + rl;
            const double dr = drs[m][reg];
            const double v = po_kt_value((double)g[m][reg], dr, dc, po_kt_rs(dr), rsc);
            if (c_ok && rr >= A.row_begin && rr < n_rows) po_out_store(&out[(rr - A.row_begin) * A.ld_out + (c - A.col_begin)], (OUT)v);
            if (mirror) wl[lr * 33 + rl] = v;
        }
        if (mirror) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const uint32_t jr = it * 2 + lh;
                const double w = wl[jr * 33 + lr];
                const uint64_t cm = cj + jr, rr = ri + m * 32 + lr;
                if (cm >= A.col_begin && cm < n_cols && rr >= A.row_begin && rr < n_rows)
                    po_out_store(&mir[(cm - A.col_begin) * A.ld_mirror + (rr - A.row_begin)], (OUT)w);
            }
        }
    }
}

}  // namespace

// ranks must fit int16 differences: dim <= 16384 (k <= 7); int32 S: dim (dim - 1) / 2 * 4 < 2^31
bool po_kt_panel_supported(uint32_t dim) { return dim > 256 && dim <= 16384 && dim % PW == 0; }
bool po_kt_panel_fold_supported(uint32_t dim, uint32_t n_selfs) { return po_kt_panel_supported(dim) && n_selfs % PW == 0; }

size_t po_kt_panel_workspace(uint64_t n, uint32_t dim) {
    const uint64_t npad = po_round_up(n ? n : 1, 128);
    return npad * (size_t)po_round_up(dim, PW) * sizeof(uint16_t) + 4096;
}

// ws layout: rank16[npad][row_words] | diagonal item list
int po_launch_kt_panel_prep(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, uint64_t npad, void* ws,
                            const uint32_t* fold_src, uint32_t fold_src_len, uint32_t n_selfs, uint32_t n_pairs,
                            po_kt_panel_plan* plan) {
    const uint32_t words = fold_src ? n_selfs + n_pairs : dim;
    const uint32_t row_words = (uint32_t)po_round_up(words, PW);
    uint16_t* rank16 = static_cast<uint16_t*>(ws);
    uint16_t* d_items = reinterpret_cast<uint16_t*>(static_cast<uint8_t*>(ws) + ((npad * (size_t)row_words * 2 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(rank16_kernel, dim3(2048), dim3(256), 0, ctx->stream, d_lessrank, n, dim, npad, fold_src, fold_src_len,
                       row_words, rank16);
    PO_CHECK_LAUNCH("rank16_kernel");
    // items of a diagonal panel pair: (p, block of 16 q) with some q > p inside the 64-word panel; all flagged partial
    // (the kernel masks q <= p and words beyond the row), padded with fully masked items to whole rounds
    static thread_local uint16_t host_items[PW * 4 + 16];
    uint32_t cnt = 0;
    for (uint32_t p = 0; p + 1 < PW; ++p)
        for (uint32_t qb = (p + 1) / 16; qb < PW / 16; ++qb) host_items[cnt++] = (uint16_t)((p << 8) | 0x80u | qb);
    while (cnt % (2 * KS)) host_items[cnt++] = (uint16_t)(((PW - 1) << 8) | 0x80u);      // p = 63, block 0: nothing survives
    PO_HIP(hipMemcpyAsync(d_items, host_items, cnt * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    plan->n_diag_items = cnt;
    plan->words = words;
    plan->row_words = row_words;
    plan->self_panels = fold_src ? n_selfs / PW : 0;
    plan->folded = fold_src ? 1 : 0;
    return PO_OK;
}

int po_launch_kt_panel_tiles(po_ctx* ctx, const po_tile_args& a, const void* ws, const po_kt_panel_plan& plan, uint64_t* tiles) {
    const uint16_t* rank16 = static_cast<const uint16_t*>(ws);
    const uint16_t* d_items = reinterpret_cast<const uint16_t*>(static_cast<const uint8_t*>(ws) + ((a.npad * (size_t)plan.row_words * 2 + 255) & ~(size_t)255));
    const uint64_t nblocks = po_tile_count(a, TM);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = 2 * 256 * kRankStride + 2 * 256 * kSigStride + ((plan.n_diag_items * 2 + 15) & ~(size_t)15);
    if (a.out_f32) {
        auto k = kt_panel_tile_kernel<float>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, rank16, plan.row_words, plan.words,
                           a.dim, plan.self_panels, plan.folded, d_items, plan.n_diag_items);
    } else {
        auto k = kt_panel_tile_kernel<double>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, rank16, plan.row_words, plan.words,
                           a.dim, plan.self_panels, plan.folded, d_items, plan.n_diag_items);
    }
    PO_CHECK_LAUNCH("kt_panel_tile_kernel");
    return PO_OK;
}
