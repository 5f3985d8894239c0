// Stage 2, exact small-integer Gram matrices over a MATERIALISED operand on the matrix cores:
//   Kendall tau   (phylodist.KT, /root/reference/phylopackage/core/phylodist.py:71-74)
//   Bray-Curtis   ('braycurtis' at /root/reference/phylopackage/bin/phyloligo.py:381) at large word spaces.
//
// Both reduce, per record pair, to one dot product of vectors with entries in {-1, 0, +1}:
//   KT:  S = sum_{p<q} sgn(x_p - x_q) sgn(y_p - y_q) = <sigma(x), sigma(y)>, sigma = the record's PAIR-SIGN vector
//        (po_kt.hip has the tie algebra that turns S into tau);
//   BC:  sum_w |a_w - b_w| = s_a + s_b - 2 sum_w min(a_w, b_w)  and  min(x, y) = sum_{t>=1} [x >= t][y >= t],
//        so sum_w min = <theta(a), theta(b)> with theta = the record's THERMOMETER code (word w owns as many 0/1
//        planes as its largest count over all records).
// Round 1 re-expanded the sign vectors inside every tile that touched a record (N/128 times per record), which made
// Kendall VALU-bound (45 ms at N = 50 000).  Here every record's vector is expanded ONCE into HBM (11 KB per record
// for folded k = 4 as int8, half that as FP4: 0.2-0.5 GB at N = 50 000 on a 288 GB part) and the N x N matrix is
// a plain symmetric Gram over it: 256 x 256 record tiles, operands staged by LDS-DMA, 128 x 64 outputs per wave.
//
// Operand layout: op[chunk][record][16 B], chunk = 16 consecutive K elements as int8 or 32 as FP4 (E2M1: 0 -> 0x0,
// +1 -> 0x2, -1 -> 0xA).  A tile's 64-record pieces of one chunk are contiguous 1 KiB runs -> one LDS-DMA
// instruction each; the 16-byte MFMA operand reads of consecutive records are conflict free.  The dot product does
// not care how K is ordered as long as both operands agree, and the A and B operand maps of one MFMA shape are the
// same function of (row | column, k), so a Gram matrix needs no knowledge of the k order inside a chunk.
//   int8: v_mfma_i32_32x32x32_i8 (int32 sums, exact);
//   FP4:  v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales (E8M0 127): twice the K per instruction at the same
//         operand bytes and cycles; products and partial sums are small integers, exact in float32 below 2^24.
// Weights: reverse-complement folded operands (po_fold.hip) give K elements of weight 4, 2, 1 (KT) or 2, 1 (BC);
// elements are ordered by class, every class is padded to whole stages, and the accumulators are doubled where a
// class ends (4 S4 + 2 S2 + S1), exact in both formats.
#include "po_tiles.h"

#include <vector>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int TE = 256;                             // tile edge (records): the unit of the tile order and of the operand padding
constexpr int TR = 256;                             // rows per workgroup (128: a tile is split between two workgroups)
constexpr int kHalves = TE / TR;
constexpr int kWaves = 4 * (TR / 128);              // each wave 128 x 64 pairs: TR / 128 rows of 4 waves
constexpr int kThreads = 64 * kWaves;
constexpr int SCH = 4;                              // 16-byte K-chunks per stage = 2 MFMA k-steps
constexpr int kChunkRow = (TR + TE) * 16;           // one chunk of the half tile's 128 row + 256 column records (6 KiB)
constexpr int kStageBytes = SCH * kChunkRow;        // 24 KiB
constexpr int NBUF = 3;                             // stage ring: two stages in flight beside the one being consumed
constexpr int kPieces = SCH * (TR + TE) / 64;       // 1 KiB LDS-DMA instructions per stage
constexpr int kTrStride = 33;
constexpr int kMirrorBytes = kWaves * 32 * kTrStride * 8;   // wave-private 32 x 32 transposes of the epilogue
constexpr int kPerWave = kPieces / kWaves;          // LDS-DMA instructions per wave and stage
static_assert(kPieces % kWaves == 0, "pieces are dealt evenly to the waves");
constexpr int kTermBytes = 2 * (TR + TE) * 8;       // two per-record terms for the half tile's rows and columns

enum { FMT_I8 = 0, FMT_FP4 = 1 };
enum { EPI_KT = 0, EPI_BC = 1 };

__host__ __device__ constexpr int elems_per_chunk(int fmt) { return fmt == FMT_I8 ? 16 : 32; }

template <int FMT> struct acc_t;
template <> struct acc_t<FMT_I8> { typedef v16i type; };
template <> struct acc_t<FMT_FP4> { typedef v16f type; };

struct pd_epilogue {
    const double* term0;     // KT: tied word pairs of every record            BC: sum of the frequencies of every record
    const double* term1;     // KT: unused                                     BC: sum of the counts (all words) of every record
    double scalar;           // KT: D (D - 1) / 2                              BC: 1 / n (n = the common word total)
};

// ---- the tile kernel -----------------------------------------------------------------------------------------
// Workgroup b: logical position L in the XCD-aware order of 2 x (number of 256 x 256 tiles); tile L / 2, rows
// [128 (L & 1), +128) of it.  The two halves of a tile are neighbours in that order: same XCD, same time, so the
// column operand they share comes out of the L2 once.
template <int FMT, int EPI, typename OUT>
__global__ __launch_bounds__(kThreads, 2) void pairdot_tile_kernel(po_tile_args A, const uint8_t* __restrict__ op,
                                                                   uint64_t op_n, uint32_t n_stages, uint32_t dbl1,
                                                                   uint32_t dbl2, pd_epilogue E) {
    typedef typename acc_t<FMT>::type ACC;
    extern __shared__ __align__(16) unsigned char smem[];   // ring [NBUF][SCH][rows 128 | cols 256][16 B]; epilogue scratch afterwards
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t wave = wv & 3, wr = wv >> 2;            // its 64-column block, its 128-row block
    const uint32_t lr = lane & 31, lh = lane >> 5;
    uint32_t ti, tj;
    const uint64_t L = po_xcd_swizzle(blockIdx.x, gridDim.x);
    po_tile_coords_logical(A, TE, L / kHalves, ti, tj);
    const uint64_t i0 = (uint64_t)ti * TE + (L % kHalves) * TR, j0 = (uint64_t)tj * TE;
    if (i0 >= min(A.row_end, A.n) || i0 + TR <= A.row_begin) return;     // a half with no row of the block (ragged edges)

    ACC g[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int e = 0; e < 16; ++e) g[m][nn][e] = 0;

    // a stage is kPieces one-KiB LDS-DMA instructions (per chunk: 2 x 64 row records, 4 x 64 column records), dealt to the waves
    auto issue = [&](uint32_t stage) {
        unsigned char* dst = smem + (stage % NBUF) * kStageBytes;
#pragma unroll
        for (int u = 0; u < kPerWave; ++u) {
            constexpr uint32_t ppc = (TR + TE) / 64;                    // pieces per chunk: TR / 64 of rows, 4 of columns
            const uint32_t idx = wv * kPerWave + u;                     // wave uniform
            const uint32_t ch = idx / ppc, p = idx % ppc;
            const uint64_t rec = (p < TR / 64 ? i0 + p * 64 : j0 + (p - TR / 64) * 64) + lane;
            po_glds16(op + ((uint64_t)(stage * SCH + ch) * op_n + rec) * 16, dst + ch * kChunkRow + p * 1024);
        }
    };
    auto compute = [&](uint32_t buf) {
#pragma unroll
        for (int s = 0; s < SCH / 2; ++s) {
            const unsigned char* base = smem + buf * kStageBytes + (2 * s + lh) * kChunkRow;   // lane halves: the two chunks of a k-step
            v4i a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4i*>(base + (wr * 128 + m * 32 + lr) * 16);
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) b[nn] = *reinterpret_cast<const v4i*>(base + (TR + wave * 64 + nn * 32 + lr) * 16);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    if constexpr (FMT == FMT_I8) {
                        g[m][nn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[nn], g[m][nn], 0, 0, 0);
                    } else {
                        const v8i a8 = {a[m][0], a[m][1], a[m][2], a[m][3], 0, 0, 0, 0};
                        const v8i b8 = {b[nn][0], b[nn][1], b[nn][2], b[nn][3], 0, 0, 0, 0};
                        // cbsz = blgp = 4: both operands FP4 (E2M1); scales E8M0 127 = 1.0 in every byte
                        g[m][nn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, g[m][nn], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    }
                }
        }
    };
    auto double_sums = [&]() {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int e = 0; e < 16; ++e) g[m][nn][e] = g[m][nn][e] + g[m][nn][e];
    };

    // Stage st + 2 is issued while stage st is consumed: a stage of two k-steps is ~0.25-0.5 us of matrix-core work,
    // an LDS-DMA from L2 / Infinity Cache takes longer than that under load, so one stage of look-ahead left the waves
    // parked at the barrier for more than half of the loop (SQ_WAIT_ANY 55 %).  A wave waits for its OWN LDS-DMA with a
    // counted vmcnt (kPieces / 4 = 6 instructions per stage: the younger stage stays in flight), then one raw s_barrier
    // per stage says that every wave's part of stage st has landed and that stage st - 1 has been consumed by all.
    static_assert(kPerWave == 6 || kPerWave == 4, "the counted waits below know 6 and 4 LDS-DMA instructions per wave and stage");
    if (n_stages > 0) issue(0);
    if (n_stages > 1) issue(1);
    for (uint32_t st = 0; st < n_stages; ++st) {
        if (st + 1 >= n_stages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kPerWave == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (st + 2 < n_stages) issue(st + 2);              // into the slot stage st - 1 has just left
        if (st == dbl1) double_sums();
        if (st == dbl2) double_sums();
        compute(st % NBUF);
    }
    __syncthreads();                                       // the ring becomes the epilogue's scratch
    if (dbl1 != PO_NO_DOUBLING && dbl1 >= n_stages) double_sums();
    if (dbl2 != PO_NO_DOUBLING && dbl2 >= n_stages) double_sums();

    // ---- epilogue --------------------------------------------------------------------------------------------
    // per-record terms into LDS first: on gfx9 loads and stores share the in-order vmcnt, a global load issued among
    // the output stores could only be waited for together with every store before it
    double* terms = reinterpret_cast<double*>(smem + kMirrorBytes);     // [term0: rows 128 | cols 256][term1: rows | cols]
    for (uint32_t x = t; x < TR + TE; x += kThreads) {
        const uint64_t rec = min((x < TR) ? i0 + x : j0 + (x - TR), A.npad - 1);
        if (EPI == EPI_KT) {                 // d = T - ties (word pairs not tied in the record) and 1/sqrt(d), once per record
            const double d = E.scalar - E.term0[rec];
            terms[x] = d;
            terms[TR + TE + x] = po_kt_rs(d);
        } else {
            terms[x] = E.term0[rec];
            terms[TR + TE + x] = E.term1[rec];
        }
    }
    __syncthreads();
    OUT* out = static_cast<OUT*>(A.out);
    OUT* mir = static_cast<OUT*>(A.mirror);
    const bool mirror = po_tile_mirrors(A, ti, tj);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const uint64_t iw = i0 + wr * 128, jw = j0 + wave * 64;
    double* wl = reinterpret_cast<double*>(smem) + wv * (32 * kTrStride);
    const double* r0t = terms + wr * 128, *c0t = terms + TR + wave * 64;
    const double* r1t = terms + TR + TE + wr * 128, *c1t = terms + TR + TE + TR + wave * 64;
    // BC: every record of a profile matrix has sum f = 1 exactly, the denominator is 2 and the division a scaling by 1/2 with the same
    // bits (po_bc_sad.hip); checked per wave on its 128 row and 64 column terms
    bool halve = false;
    if (EPI != EPI_KT) {
        const uint32_t ln = lh * 32 + lr;
        halve = __builtin_amdgcn_ballot_w64(!(r0t[ln] == 1.0 && r0t[64 + ln] == 1.0 && c0t[ln] == 1.0)) == 0ull;
    }
    const double half_scalar = 0.5 * E.scalar;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const uint64_t c = jw + nn * 32 + lr;
        const bool c_ok = c >= A.col_begin && c < n_cols;
        const double tc0 = c0t[nn * 32 + lr], tc1 = c1t[nn * 32 + lr];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;    // accumulator layout: column = lane & 31
                const uint64_t r = iw + m * 32 + rl;
                const double G = (double)g[m][nn][reg];
                double v;
                if (EPI == EPI_KT) {          // po_tiles.h: tau = S * (rs_r rs_c), KT = 1 - (1 - tau), 0 when a factor vanishes
                    v = po_kt_value(G, r0t[m * 32 + rl], tc0, r1t[m * 32 + rl], tc1);
                } else {                      // BC = (sum |ca - cb| / n) / (w_a + w_b),  sum |ca - cb| = s_a + s_b - 2 sum min
                    const double num = (r1t[m * 32 + rl] + tc1) - 2.0 * G;      // exact integers
                    const double x = halve ? num * half_scalar : (num * E.scalar) / (r0t[m * 32 + rl] + tc0);
                    v = (r == c) ? 0.0 : x;
                }
                if (c_ok && r >= A.row_begin && r < n_rows) po_out_store(&out[(r - A.row_begin) * A.ld_out + (c - A.col_begin)], (OUT)v);
                if (mirror) wl[lr * kTrStride + rl] = v;
            }
            if (mirror) {                                  // wave-private scratch; LDS operations of a wave run in order
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const uint32_t jr = it * 2 + lh;
                    const double w = wl[jr * kTrStride + lr];
                    const uint64_t cm = jw + nn * 32 + jr, r = iw + m * 32 + lr;
                    if (cm >= A.col_begin && cm < n_cols && r >= A.row_begin && r < n_rows)
                        po_out_store(&mir[(cm - A.col_begin) * A.ld_mirror + (r - A.row_begin)], (OUT)w);
                }
            }
        }
    }
}

// ---- Kendall: rank rows -> pair-sign operand ---------------------------------------------------------------
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

// One workgroup = 64 records (a wave's lanes), their rank rows transposed in LDS; the four waves walk the chunks.
// pq[k] = p | q << 16: element k of the operand is sgn(x_q - x_p); padding elements have p == q (sign 0).
template <int FMT>
__global__ __launch_bounds__(256) void kt_expand_kernel(const uint8_t* __restrict__ rank8, uint32_t row_bytes, uint64_t rank_rows,
                                                        const uint32_t* __restrict__ pq, uint32_t n_chunks,
                                                        uint8_t* __restrict__ op, uint64_t op_n) {
    extern __shared__ __align__(16) unsigned char rT[];    // [row_bytes][64]
    constexpr int EPC = elems_per_chunk(FMT);
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint64_t rec0 = (uint64_t)blockIdx.x * 64;
    for (uint32_t idx = t; idx < 64 * row_bytes; idx += 256) {
        const uint32_t r = idx / row_bytes, c = idx - r * row_bytes;
        rT[c * 64 + r] = (rec0 + r < rank_rows) ? rank8[(rec0 + r) * row_bytes + c] : (uint8_t)0;
    }
    __syncthreads();
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (t >> 6));
    for (uint32_t ch = w0; ch < n_chunks; ch += gridDim.y * 4) {
        const uint32_t* codes = pq + (size_t)ch * EPC;      // wave uniform
        uint32_t word[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const uint32_t code = codes[e];
            const int a = rT[(code & 0xFFFFu) * 64 + lane], b = rT[(code >> 16) * 64 + lane];
            const int s = (b > a) - (b < a);
            if (FMT == FMT_I8) word[e >> 2] |= ((uint32_t)s & 0xFFu) << (8 * (e & 3));
            else word[e >> 3] |= (s > 0 ? 0x2u : (s < 0 ? 0xAu : 0x0u)) << (4 * (e & 7));
        }
        *reinterpret_cast<uint4*>(op + ((uint64_t)ch * op_n + rec0 + lane) * 16) = make_uint4(word[0], word[1], word[2], word[3]);
    }
}

// ---- word spaces above 256 words: uint16 ranks, kept transposed in HBM (an LDS copy of 64 rank rows no longer fits) ----
// rankT[c][rec] = lessrank[rec][src ? src[c] : c]  (uint16; zero beyond the words / records)
__global__ __launch_bounds__(256) void rank16t_kernel(const uint32_t* __restrict__ lessrank, uint64_t n, uint32_t dim,
                                                      const uint32_t* __restrict__ src, uint32_t words, uint64_t op_n,
                                                      uint16_t* __restrict__ rankT) {
    const uint64_t total = (uint64_t)words * op_n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t c = i / op_n, r = i - c * op_n;
        const uint32_t w = src ? src[c] : (uint32_t)c;                 // 0xFFFFFFFF: padding column
        rankT[i] = (r < n && w < dim) ? (uint16_t)lessrank[r * dim + w] : (uint16_t)0;
    }
}

// thread = (record, chunk).  The pair table walks q for a fixed p, so x_p stays in a register for most elements.
template <int FMT>
__global__ __launch_bounds__(256) void kt_expand16_kernel(const uint16_t* __restrict__ rankT, const uint32_t* __restrict__ pq,
                                                          uint32_t n_chunks, uint8_t* __restrict__ op, uint64_t op_n) {
    constexpr int EPC = elems_per_chunk(FMT);
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint64_t rec = (uint64_t)blockIdx.x * 64 + lane;             // < op_n: the grid covers the padded records
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (t >> 6));
    for (uint32_t ch = w0; ch < n_chunks; ch += gridDim.y * 4) {
        const uint32_t* codes = pq + (size_t)ch * EPC;      // wave uniform
        uint32_t word[4] = {0, 0, 0, 0};
        uint32_t cached_p = 0xFFFFFFFFu;
        int a = 0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const uint32_t code = codes[e];
            const uint32_t p = code & 0xFFFFu, q = code >> 16;
            if (p != cached_p) {                             // uniform branch
                cached_p = p;
                a = rankT[(uint64_t)p * op_n + rec];
            }
            const int b = rankT[(uint64_t)q * op_n + rec];
            const int s = (b > a) - (b < a);
            if (FMT == FMT_I8) word[e >> 2] |= ((uint32_t)s & 0xFFu) << (8 * (e & 3));
            else word[e >> 3] |= (s > 0 ? 0x2u : (s < 0 ? 0xAu : 0x0u)) << (4 * (e & 7));
        }
        *reinterpret_cast<uint4*>(op + ((uint64_t)ch * op_n + rec) * 16) = make_uint4(word[0], word[1], word[2], word[3]);
    }
}

template <int FMT, int EPI>
int launch_tiles(po_ctx* ctx, const po_tile_args& a, const uint8_t* op, uint64_t op_n, uint32_t n_stages, uint32_t dbl1,
                 uint32_t dbl2, const pd_epilogue& E, uint64_t* tiles) {
    const uint64_t nblocks = kHalves * po_tile_count(a, TE);       // kHalves workgroups (row blocks of TR records) per tile
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    const size_t shmem = NBUF * (size_t)kStageBytes;
    static_assert(kMirrorBytes + kTermBytes <= NBUF * kStageBytes, "epilogue scratch must fit the staging area");
    if (a.out_f32) {
        auto k = pairdot_tile_kernel<FMT, EPI, float>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, op, op_n, n_stages, dbl1, dbl2, E);
    } else {
        auto k = pairdot_tile_kernel<FMT, EPI, double>;
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)nblocks), dim3(kThreads), shmem, ctx->stream, a, op, op_n, n_stages, dbl1, dbl2, E);
    }
    PO_CHECK_LAUNCH("pairdot_tile_kernel");
    return PO_OK;
}

}  // namespace

// ---- Kendall ---------------------------------------------------------------------------------------------------
bool po_kt_pairdot_supported(uint32_t dim) { return dim >= 2 && dim <= 16384; }      // uint8 ranks up to 256 words, uint16 above

static uint32_t kt_classes(uint32_t words, uint32_t n_selfs, bool folded, uint64_t cnt[3]) {
    if (!folded) { cnt[0] = (uint64_t)words * (words - 1) / 2; cnt[1] = cnt[2] = 0; return 1; }
    const uint64_t reps = words - n_selfs;
    cnt[0] = reps * (reps - 1) / 2;                 // weight 4: both words stand for two-word orbits
    cnt[1] = reps * n_selfs;                        // weight 2
    cnt[2] = (uint64_t)n_selfs * (n_selfs ? n_selfs - 1 : 0) / 2;   // weight 1
    return 3;
}

// float32 sums of FP4 products are exact below 2^24: |S| <= D (D - 1) / 2 decides the operand format
static bool kt_fp4_exact(uint32_t dim) { return (uint64_t)dim * (dim - 1) / 2 < (1ull << 24); }

// rank buffer: uint8 [npad][round16(D)] (D <= 256) or uint16 transposed [D][round256(n)]
size_t po_kt_pairdot_rank_bytes(uint64_t n, uint32_t dim) {
    if (dim <= 256) return po_round_up(n ? n : 1, 128) * po_round_up(dim, 16) + 256;
    return (size_t)dim * po_round_up(n ? n : 1, TE) * sizeof(uint16_t) + 256;
}

// bytes of the materialised operand for n records of `words` kept words (n_selfs of them self-paired when folded)
size_t po_kt_pairdot_operand_bytes(uint64_t n, uint32_t dim, uint32_t words, uint32_t n_selfs, bool folded, int fmt_fp4) {
    const uint64_t epc = (fmt_fp4 && kt_fp4_exact(dim)) ? 32 : 16, per_stage = epc * SCH;
    uint64_t cnt[3], k = 0;
    const uint32_t nc = kt_classes(words, n_selfs, folded, cnt);
    for (uint32_t c = 0; c < nc; ++c) k += po_round_up(cnt[c], per_stage);
    if (k == 0) k = per_stage;
    return (k / epc) * po_round_up(n ? n : 1, TE) * 16 + 256;
}

// Builds the pair table (cached in the context per (dim, fold layout, format)), the uint8 rank rows and the operand.
int po_launch_kt_pairdot_prep(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, uint64_t npad,
                              const uint32_t* fold_src, uint32_t n_selfs, uint32_t n_pairs, int fmt_fp4,
                              po_pairdot_plan* plan) {
    const bool folded = fold_src != nullptr;
    const uint32_t words = folded ? n_selfs + n_pairs : dim;
    const uint32_t row_bytes = folded ? (uint32_t)po_round_up(words, 16) : dim;
    if (!kt_fp4_exact(dim)) fmt_fp4 = 0;                   // k = 7: sums up to 1.3e8 need the int32 accumulators
    const uint32_t epc = fmt_fp4 ? 32u : 16u, per_stage = epc * SCH;
    uint64_t cnt[3];
    const uint32_t n_classes = kt_classes(words, n_selfs, folded, cnt);
    uint64_t kpad = 0;
    uint32_t class_stage[4] = {0, 0, 0, 0};
    for (uint32_t c = 0; c < n_classes; ++c) {
        class_stage[c] = (uint32_t)(kpad / per_stage);
        kpad += po_round_up(cnt[c], per_stage);
    }
    if (kpad == 0) kpad = per_stage;
    class_stage[n_classes] = (uint32_t)(kpad / per_stage);
    const uint64_t key = ((uint64_t)dim << 32) | ((uint64_t)(folded ? n_selfs + 1 : 0) << 8) | (uint64_t)(fmt_fp4 ? 1 : 0);
    int rc;
    if (ctx->pq_key != key || ctx->ws_pq.p == nullptr) {
        std::vector<uint32_t> pq(kpad, 0u);               // p == q == 0: a zero sign
        size_t at = 0;
        for (uint32_t c = 0; c < n_classes; ++c) {
            at = (size_t)class_stage[c] * per_stage;
            const uint32_t want = folded ? (4u >> c) : 0u;
            for (uint32_t p = 0; p + 1 < words; ++p)
                for (uint32_t q = p + 1; q < words; ++q) {
                    if (folded && (p < n_selfs ? 1u : 2u) * (q < n_selfs ? 1u : 2u) != want) continue;
                    pq[at++] = p | (q << 16);
                }
        }
        rc = po_buf_reserve(ctx, &ctx->ws_pq, kpad * sizeof(uint32_t));
        if (rc) return rc;
        PO_HIP(hipMemcpyAsync(ctx->ws_pq.p, pq.data(), kpad * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PO_HIP(hipStreamSynchronize(ctx->stream));        // pq is a local vector
        ctx->pq_key = key;
    }
    const uint64_t op_n = po_round_up(n, TE);
    const uint32_t n_chunks = (uint32_t)(kpad / epc);
    rc = po_buf_reserve(ctx, &ctx->ws_pairdot, (size_t)n_chunks * op_n * 16);
    if (rc) return rc;
    const uint32_t* pq = static_cast<const uint32_t*>(ctx->ws_pq.p);
    uint8_t* op = static_cast<uint8_t*>(ctx->ws_pairdot.p);
    if (dim <= 256) {
        uint8_t* rank8 = static_cast<uint8_t*>(ctx->ws_freq.p);        // reserved by the caller (po_kt_pairdot_rank_bytes)
        hipLaunchKernelGGL(rank8_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_lessrank, n, dim, npad, fold_src, row_bytes, rank8);
        PO_CHECK_LAUNCH("rank8_kernel");
        const dim3 grid((uint32_t)(op_n / 64), 8);
        const size_t shmem = (size_t)row_bytes * 64;
        if (fmt_fp4) hipLaunchKernelGGL(kt_expand_kernel<FMT_FP4>, grid, dim3(256), shmem, ctx->stream, rank8, row_bytes, npad, pq, n_chunks, op, op_n);
        else hipLaunchKernelGGL(kt_expand_kernel<FMT_I8>, grid, dim3(256), shmem, ctx->stream, rank8, row_bytes, npad, pq, n_chunks, op, op_n);
        PO_CHECK_LAUNCH("kt_expand_kernel");
    } else {
        uint16_t* rankT = static_cast<uint16_t*>(ctx->ws_freq.p);
        hipLaunchKernelGGL(rank16t_kernel, dim3(2048), dim3(256), 0, ctx->stream, d_lessrank, n, dim, fold_src, words, op_n, rankT);
        PO_CHECK_LAUNCH("rank16t_kernel");
        uint32_t gy = (n_chunks + 255) / 256;                          // ~64 chunks per wave
        gy = gy < 1 ? 1 : (gy > 4096 ? 4096 : gy);
        const dim3 grid((uint32_t)(op_n / 64), gy);
        if (fmt_fp4) hipLaunchKernelGGL(kt_expand16_kernel<FMT_FP4>, grid, dim3(256), 0, ctx->stream, rankT, pq, n_chunks, op, op_n);
        else hipLaunchKernelGGL(kt_expand16_kernel<FMT_I8>, grid, dim3(256), 0, ctx->stream, rankT, pq, n_chunks, op, op_n);
        PO_CHECK_LAUNCH("kt_expand16_kernel");
    }
    plan->fmt_fp4 = fmt_fp4 ? 1 : 0;
    plan->n_stages = (uint32_t)(kpad / per_stage);
    plan->op_n = op_n;
    plan->dbl1 = folded ? class_stage[1] : PO_NO_DOUBLING;             // == n_stages when the later classes are empty
    plan->dbl2 = folded ? class_stage[2] : PO_NO_DOUBLING;
    plan->k_elems = kpad;
    return PO_OK;
}

int po_launch_kt_pairdot_tiles(po_ctx* ctx, const po_tile_args& a, const po_pairdot_plan& plan, uint64_t* tiles) {
    pd_epilogue E;
    E.term0 = a.rowstat + 3 * a.npad;                                  // tied word pairs (po_launch_ranks)
    E.term1 = nullptr;
    E.scalar = 0.5 * (double)a.dim * ((double)a.dim - 1.0);
    const uint8_t* op = static_cast<const uint8_t*>(ctx->ws_pairdot.p);
    return plan.fmt_fp4 ? launch_tiles<FMT_FP4, EPI_KT>(ctx, a, op, plan.op_n, plan.n_stages, plan.dbl1, plan.dbl2, E, tiles)
                        : launch_tiles<FMT_I8, EPI_KT>(ctx, a, op, plan.op_n, plan.n_stages, plan.dbl1, plan.dbl2, E, tiles);
}

// ---- Bray-Curtis: thermometer planes ------------------------------------------------------------------------------
// Eligible when the packed-byte prep of po_bc_sad.hip says so for the WHOLE matrix: every record has the same word
// total n and every count is <= 255 (fixed-length contigs, windows, reads - the equal-total regime of the SAD
// kernel).  Then  BC(a, b) = (s_a + s_b - 2 sum_w min(ca_w, cb_w)) / n / (w_a + w_b)  and sum_w min is the dot
// product of the records' thermometer codes: word w owns T_w = max_r c_{r,w} planes, plane t holds [c_w >= t].
// With 4^6 words and 2 kb contigs T_w is 7-9 although the byte range would allow 255: the operand is
// dim_f x ~8 zero/one elements per record (FP4: 8 KB), and the SAD kernel's 1/4 VALU instruction per word and pair
// becomes a matrix-core Gram.  Folded operands (po_fold.hip): representatives first (weight 2), self-paired words
// after (weight 1), the accumulators double once in between.
namespace {

struct thermo_header {          // device + pinned host copy
    uint32_t k0_stages;         // stages of the first class (== total stages when there is one class)
    uint32_t k_stages;          // total stages
    uint32_t ok;                // 1: every 128-record block qualifies and all blocks share one total
    uint32_t pad;
    unsigned long long ntot;    // the common word total
    unsigned long long k_raw;   // planes before padding (reporting)
};

// colmax[w] = largest count of word w over all records, from the packed transposed matrix P8t[g][npad]
__global__ __launch_bounds__(256) void bc_colmax_kernel(const uint32_t* __restrict__ p8t, uint64_t n, uint64_t npad,
                                                        uint32_t* __restrict__ colmax4) {
    __shared__ uint32_t red[4][4];
    const uint32_t g = blockIdx.x;
    const uint32_t* row = p8t + (uint64_t)g * npad;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (uint64_t r = threadIdx.x; r < n; r += 256) {
        const uint32_t v = row[r];
        m0 = max(m0, v & 255u); m1 = max(m1, (v >> 8) & 255u); m2 = max(m2, (v >> 16) & 255u); m3 = max(m3, v >> 24);
    }
    for (int o = 32; o > 0; o >>= 1) {
        m0 = max(m0, (uint32_t)__shfl_down(m0, o, 64)); m1 = max(m1, (uint32_t)__shfl_down(m1, o, 64));
        m2 = max(m2, (uint32_t)__shfl_down(m2, o, 64)); m3 = max(m3, (uint32_t)__shfl_down(m3, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = m0; red[threadIdx.x >> 6][1] = m1; red[threadIdx.x >> 6][2] = m2; red[threadIdx.x >> 6][3] = m3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t e = threadIdx.x;
        colmax4[g * 4 + e] = max(max(red[0][e], red[1][e]), max(red[2][e], red[3][e]));
    }
}

// One workgroup: element offsets of every word (class 0 = words below dbl_at, class 1 = the rest; each class padded
// to whole stages of `per_stage` elements) and the eligibility of the matrix.
__global__ __launch_bounds__(1024) void bc_plan_kernel(const uint32_t* __restrict__ colmax, uint32_t words, uint32_t dbl_at,
                                                       uint32_t per_stage, const unsigned long long* __restrict__ cls,
                                                       uint32_t n_cls, uint32_t* __restrict__ off, thermo_header* __restrict__ hdr) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long base_s;
    __shared__ int ok_s;
    const uint32_t t = threadIdx.x;
    if (t == 0) { base_s = 0; ok_s = 1; }
    __syncthreads();
    const unsigned long long ref = n_cls ? cls[0] : 0ull;
    bool ok = ref != 0ull;
    for (uint32_t b = t; b < n_cls; b += 1024) ok = ok && cls[b] == ref;
    if (!ok) ok_s = 0;
    const uint32_t split = dbl_at < words ? dbl_at : words;
    unsigned long long k0 = 0;
    for (int cl = 0; cl < 2; ++cl) {
        const uint32_t lo = cl == 0 ? 0 : split, hi = cl == 0 ? split : words;
        for (uint32_t w0 = lo; w0 < hi; w0 += 1024) {
            const uint32_t w = w0 + t;
            const unsigned long long v = (w < hi) ? colmax[w] : 0ull;
            part[t] = v;
            __syncthreads();
            for (uint32_t s = 1; s < 1024; s <<= 1) {          // inclusive scan
                const unsigned long long add = t >= s ? part[t - s] : 0ull;
                __syncthreads();
                part[t] += add;
                __syncthreads();
            }
            if (w < hi) off[w] = (uint32_t)(base_s + part[t] - v);
            __syncthreads();
            if (t == 1023) base_s += part[1023];
            __syncthreads();
        }
        if (t == 0) {
            const unsigned long long padded = (base_s + per_stage - 1) / per_stage * per_stage;
            if (cl == 0) { k0 = padded; hdr->k0_stages = (uint32_t)(padded / per_stage); hdr->k_raw = base_s; }
            else { hdr->k_stages = (uint32_t)(padded / per_stage); hdr->k_raw += base_s - k0; }
            base_s = padded;
        }
        __syncthreads();
    }
    if (t == 0) {
        hdr->ok = ok_s ? 1u : 0u;
        hdr->ntot = ref;
        hdr->pad = 0;
    }
}

// emap[k] = t << 20 | w for plane t (1-based) of word w; untouched elements stay 0xFFFFFFFF (padding: value 0)
__global__ __launch_bounds__(256) void bc_emap_kernel(const uint32_t* __restrict__ colmax, const uint32_t* __restrict__ off,
                                                      uint32_t words, uint32_t* __restrict__ emap) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    const uint32_t T = colmax[w], o = off[w];
    for (uint32_t t = 1; t <= T; ++t) emap[o + t - 1] = (t << 20) | w;
}

// s[r] = sum of the counts of record r over ALL words (folded: representatives count twice).  64 records per workgroup, four lanes
// per record each walking every fourth word group (one thread per record walked 520 groups in a row at k = 6, with 196 workgroups
// on 256 CUs: 140 us for 104 MB).
__global__ __launch_bounds__(256) void bc_rowsum_kernel(const uint32_t* __restrict__ p8t, uint32_t groups, uint32_t dbl_group,
                                                        uint64_t n, uint64_t npad, uint64_t out_n, double* __restrict__ s) {
    __shared__ unsigned long long part[2][4][64];
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * 64 + tx;
    unsigned long long a = 0, b = 0;
    if (r < n) {
        for (uint32_t g = ty; g < groups; g += 4) {
            const uint32_t v = p8t[(uint64_t)g * npad + r];
            const uint32_t sum = (v & 255u) + ((v >> 8) & 255u) + ((v >> 16) & 255u) + (v >> 24);
            if (g < dbl_group) a += sum; else b += sum;
        }
    }
    part[0][ty][tx] = a;
    part[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && r < out_n) {
        a = part[0][0][tx] + part[0][1][tx] + part[0][2][tx] + part[0][3][tx];
        b = part[1][0][tx] + part[1][1][tx] + part[1][2][tx] + part[1][3][tx];
        s[r] = (double)(2ull * a + b);
    }
}

// thread = (record, chunk): EPC thermometer bits of one record
template <int FMT>
__global__ __launch_bounds__(256) void bc_expand_kernel(const uint32_t* __restrict__ p8t, uint64_t n, uint64_t npad,
                                                        const uint32_t* __restrict__ emap, uint32_t n_chunks,
                                                        uint8_t* __restrict__ op, uint64_t op_n) {
    constexpr int EPC = elems_per_chunk(FMT);
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint64_t rec = (uint64_t)blockIdx.x * 64 + lane;
    const bool live = rec < n;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (t >> 6));
    for (uint32_t ch = w0; ch < n_chunks; ch += gridDim.y * 4) {
        const uint32_t* codes = emap + (size_t)ch * EPC;    // wave uniform
        uint32_t word[4] = {0, 0, 0, 0};
        uint32_t cached_g = 0xFFFFFFFFu, cached_v = 0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const uint32_t code = codes[e];
            if (code == 0xFFFFFFFFu) continue;              // padding (uniform branch)
            const uint32_t w = code & 0xFFFFFu, lvl = code >> 20;
            if ((w >> 2) != cached_g) {                     // uniform: consecutive planes mostly stay in one word group
                cached_g = w >> 2;
                cached_v = live ? p8t[(uint64_t)cached_g * npad + rec] : 0u;
            }
            const uint32_t c = (cached_v >> (8 * (w & 3))) & 255u;
            const uint32_t bit = c >= lvl ? 1u : 0u;
            if (FMT == FMT_I8) word[e >> 2] |= bit << (8 * (e & 3));
            else word[e >> 3] |= (bit << 1) << (4 * (e & 7));           // FP4 1.0 = 0x2
        }
        *reinterpret_cast<uint4*>(op + ((uint64_t)ch * op_n + rec) * 16) = make_uint4(word[0], word[1], word[2], word[3]);
    }
}

}  // namespace

// p8t / cls: the packed transposed counts and block classes of po_launch_bc_sad_prep (same workspace).
// *eligible = false: leave the matrix to the SAD / general kernels.  Reads one small header back (one stream sync).
int po_launch_bc_thermo_prep(po_ctx* ctx, const uint32_t* p8t, uint32_t groups_pad, const unsigned long long* cls, uint64_t n,
                             uint32_t dim, uint64_t npad, uint32_t dbl_at, int fmt_fp4, bool* eligible, po_pairdot_plan* plan,
                             double* inv_n) {
    *eligible = false;
    const uint32_t epc = fmt_fp4 ? 32u : 16u, per_stage = epc * SCH;
    const uint32_t words = groups_pad * 4;                 // padded word columns are all zero: no planes
    if (words >= (1u << 20)) return PO_OK;                 // emap packs the word index into 20 bits (bc_emap_kernel)
    const uint64_t op_n = po_round_up(n, TE);
    // ws_thermo: header | colmax[words] | off[words] | rowsum[op_n] | emap[...]
    const size_t b_hdr = 256, b_cm = po_round_up((size_t)words * 4, 256), b_off = b_cm, b_rs = po_round_up(op_n * sizeof(double), 256);
    int rc = po_buf_reserve(ctx, &ctx->ws_thermo, b_hdr + b_cm + b_off + b_rs);
    if (rc) return rc;
    if (!ctx->h_flag) PO_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_flag), 64, hipHostMallocDefault));
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_thermo.p);
    thermo_header* hdr = reinterpret_cast<thermo_header*>(base);
    uint32_t* colmax = reinterpret_cast<uint32_t*>(base + b_hdr);
    uint32_t* off = reinterpret_cast<uint32_t*>(base + b_hdr + b_cm);
    hipLaunchKernelGGL(bc_colmax_kernel, dim3(groups_pad), dim3(256), 0, ctx->stream, p8t, n, npad, colmax);
    PO_CHECK_LAUNCH("bc_colmax_kernel");
    hipLaunchKernelGGL(bc_plan_kernel, dim3(1), dim3(1024), 0, ctx->stream, colmax, words, dbl_at, per_stage, cls,
                       (uint32_t)((n + 127) / 128), off, hdr);
    PO_CHECK_LAUNCH("bc_plan_kernel");
    static_assert(sizeof(thermo_header) <= 64, "header must fit the pinned flag block");
    PO_HIP(hipMemcpyAsync(ctx->h_flag, hdr, sizeof(thermo_header), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    const thermo_header h = *reinterpret_cast<const thermo_header*>(ctx->h_flag);
    const uint64_t kpad = (uint64_t)h.k_stages * per_stage;
    // worth it while a word owns few planes (the matrix cores do ~27 FP4 planes in the time the SAD kernel does one
    // word), and bounded in memory
    const uint64_t op_bytes = kpad / epc * op_n * 16;
    if (!h.ok || h.k_stages == 0 || kpad > (uint64_t)(fmt_fp4 ? 24 : 12) * dim || op_bytes > PO_PAIRDOT_MAX_OPERAND) return PO_OK;
    // float32 sums of FP4 products are exact below 2^24: sum_w min(ca_w, cb_w) <= n (doubling included).  Counts <= 255 and
    // 4^8 words keep n below that, a recovered matrix of arbitrary width need not
    if (fmt_fp4 && h.ntot >= (1ull << 24)) return PO_OK;
    // emap lives behind the row sums; the buffer may move when it grows, so re-derive the pointers afterwards
    rc = po_buf_reserve(ctx, &ctx->ws_pairdot, op_bytes + kpad * sizeof(uint32_t) + 256);
    if (rc) return rc;
    uint8_t* op = static_cast<uint8_t*>(ctx->ws_pairdot.p);
    uint32_t* emap = reinterpret_cast<uint32_t*>(op + po_round_up(op_bytes, 256));
    double* rowsum = reinterpret_cast<double*>(base + b_hdr + b_cm + b_off);
    PO_HIP(hipMemsetAsync(emap, 0xFF, kpad * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(bc_emap_kernel, dim3((words + 255) / 256), dim3(256), 0, ctx->stream, colmax, off, words, emap);
    PO_CHECK_LAUNCH("bc_emap_kernel");
    const uint32_t dbl_group = dbl_at == PO_NO_DOUBLING ? 0u : dbl_at / 4;     // unfolded: every word counts once (class b)
    hipLaunchKernelGGL(bc_rowsum_kernel, dim3((uint32_t)((op_n + 63) / 64)), dim3(256), 0, ctx->stream, p8t, groups_pad,
                       dbl_group, n, npad, op_n, rowsum);
    PO_CHECK_LAUNCH("bc_rowsum_kernel");
    const uint32_t n_chunks = (uint32_t)(kpad / epc);
    const dim3 grid((uint32_t)(op_n / 64), 16);
    if (fmt_fp4) hipLaunchKernelGGL(bc_expand_kernel<FMT_FP4>, grid, dim3(256), 0, ctx->stream, p8t, n, npad, emap, n_chunks, op, op_n);
    else hipLaunchKernelGGL(bc_expand_kernel<FMT_I8>, grid, dim3(256), 0, ctx->stream, p8t, n, npad, emap, n_chunks, op, op_n);
    PO_CHECK_LAUNCH("bc_expand_kernel");
    plan->fmt_fp4 = fmt_fp4 ? 1 : 0;
    plan->n_stages = h.k_stages;
    plan->op_n = op_n;
    plan->dbl1 = dbl_at == PO_NO_DOUBLING ? PO_NO_DOUBLING : h.k0_stages;
    plan->dbl2 = PO_NO_DOUBLING;
    plan->k_elems = h.k_raw;
    plan->aux_offset = b_hdr + b_cm + b_off;                           // the row sums inside ws_thermo
    *inv_n = 1.0 / (double)h.ntot;
    *eligible = true;
    return PO_OK;
}

int po_launch_bc_thermo_tiles(po_ctx* ctx, const po_tile_args& a, const po_pairdot_plan& plan, double inv_n, uint64_t* tiles) {
    pd_epilogue E;
    E.term0 = a.rowstat + a.npad;                                      // sum of the frequencies of every record
    const uint8_t* base = static_cast<const uint8_t*>(ctx->ws_thermo.p);
    E.term1 = reinterpret_cast<const double*>(base + plan.aux_offset);
    E.scalar = inv_n;
    const uint8_t* op = static_cast<const uint8_t*>(ctx->ws_pairdot.p);
    return plan.fmt_fp4 ? launch_tiles<FMT_FP4, EPI_BC>(ctx, a, op, plan.op_n, plan.n_stages, plan.dbl1, plan.dbl2, E, tiles)
                        : launch_tiles<FMT_I8, EPI_BC>(ctx, a, op, plan.op_n, plan.n_stages, plan.dbl1, plan.dbl2, E, tiles);
}
