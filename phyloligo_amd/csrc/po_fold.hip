// Reverse-complement folding of strand-symmetric profiles (stage 2: JSD, Bray-Curtis, Kendall).
//
// Under `-s both` the reference counts the words of  seq + revcomp(seq)  (select_strand,
// /root/reference/phylopackage/bin/phyloligo.py:124-149, :141).  That string is its own reverse
// complement, so whenever the pattern reads the same in both directions (every contiguous k-mer, `-k`; spaced
// patterns such as 11011011) each window has a mirror window holding the reverse-complemented word, and
//     count[w] == count[rc(w)]   exactly, for every record.
// In the C,G,A,T digit coding of count2freq (:653) rc(w) = digits reversed, each XOR 1.  A metric that
// is a sum over words of f(a_w, b_w) then only needs one word per orbit {w, rc(w)}:
//     sum_w f(a_w, b_w) = 2 * sum_{w < rc(w)} f(a_w, b_w) + sum_{w == rc(w)} f(a_w, b_w),
// i.e. 136 instead of 256 words at k=4 and 2080 instead of 4096 at k=6.  The property is CHECKED on the device for
// the matrix at hand (one pass, fused with the fold itself) - it is a fact about the data, not a promise of
// the caller - and the folded matrix is used only if every record has it.  Layout of a folded record:
//     [ one representative per two-word orbit, ascending, zero padded to `gran` words | self-paired words, padded to 8 ]
// The tile kernels walk it front to back and double their accumulators once, at word `dbl_at`
// (po_tile_args), so the folded sums carry the same weights as the full ones.
#include "po_internal.h"

#include <vector>

namespace {

__device__ __host__ inline uint32_t rc_word(uint32_t w, uint32_t k) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < k; ++i) {
        r = (r << 2) | ((w & 3u) ^ 1u);
        w >>= 2;
    }
    return r;
}

// Could record `row` belong to a 128-record block with one common total?  Only if it shares its non-zero total with the
// records 1, 3, 7, 15, 31, 63 and 127 places on (taken cyclically inside its block; a block of one record is its own mate).
// In a block that does have one total every record says yes.  In a ragged assembly seven coincidences at once do not happen
// even when a tenth of the records share a total (contigs cut off at a minimum length: 0.1^8 per record), so that "nobody
// said yes" tells the host - in the one flag word it reads anyway - that the equal-total kernels would own no tile and need
// neither their operands nor their launches.
__device__ __forceinline__ bool fold_some_equal(const unsigned long long* __restrict__ totals, uint64_t row, uint64_t n,
                                                unsigned long long tot) {
    const uint64_t first = row & ~(uint64_t)127;
    const uint32_t size = (uint32_t)min((uint64_t)128, n - first), at = (uint32_t)(row - first);   // 32-bit: a 64-bit % is ~150 instructions
    unsigned long long mate[7];                          // seven independent loads in flight, not a chain of short-circuited ones
#pragma unroll
    for (uint32_t q = 0; q < 7; ++q) {
        const uint32_t x = at + ((2u << q) - 1u);        // < 256
        mate[q] = totals[first + (size == 128u ? (x & 127u) : x % size)];
    }
    uint32_t same = tot > 0 ? 1u : 0u;
#pragma unroll
    for (uint32_t q = 0; q < 7; ++q) same &= (mate[q] == tot) ? 1u : 0u;
    return same != 0u;
}

// A workgroup stages `rpb` whole records in LDS with coalesced loads (records are contiguous in memory), then
// every lane takes folded columns: the representative word and its reverse complement are compared (both
// reads hit LDS, so the scattered partner costs nothing in HBM) and the representative is written out.
template <typename T>
__global__ __launch_bounds__(256) void rc_fold_kernel(const T* __restrict__ in, uint64_t n, uint32_t dim, uint32_t k,
                                                      const uint32_t* __restrict__ src, uint32_t dim_f, uint32_t rpb,
                                                      T* __restrict__ out, uint32_t* __restrict__ asym,
                                                      const unsigned long long* __restrict__ totals) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long sum_s[16];
    __shared__ uint32_t max_s[16];
    T* rec = reinterpret_cast<T*>(smem);
    if (threadIdx.x < 16) { sum_s[threadIdx.x] = 0; max_s[threadIdx.x] = 0; }
    const uint64_t row0 = (uint64_t)blockIdx.x * rpb;
    const uint32_t rows = (uint32_t)min((uint64_t)rpb, n - row0);
    // Where word i of a record sits in the LDS copy.  The partner reads of a wave - rc(w) for 64 consecutive representatives w - differ in
    // their HIGH digits only (the reversed low digits of w) and would all fall into one bank: 64-way conflicts made this pass LDS-bound
    // (432 us at k = 6 whatever the arithmetic around it).  Folding the bits above the bank bits into them - a permutation inside every
    // group of 64 words - spreads both the representatives and their partners over the banks.
    auto slot = [](uint32_t i) { return i ^ ((i >> 6) & 63u) ^ ((i >> 12) & 63u); };
    // the plan's two tables come into LDS with the records: read from global memory word by word inside the fold loop they are a chain of
    // dependent L2 round trips per lane (eight at k = 6)
    // (up to 1 024 folded words; beyond, the 16 KiB they would take cost more in resident workgroups than the round trips)
    const bool tables_in_lds = dim_f <= 1024u;
    uint32_t* lsrc = reinterpret_cast<uint32_t*>(smem + (size_t)rpb * dim * sizeof(T));
    if (tables_in_lds)
        for (uint32_t e = threadIdx.x; e < 2u * dim_f; e += 256) lsrc[e] = src[e];
    const uint32_t* tsrc = tables_in_lds ? lsrc : src;
    const T* x = in + row0 * dim;
    const uint32_t dmask = dim - 1u;                       // dim = 4^k
    constexpr uint32_t per16 = 16u / (uint32_t)sizeof(T);
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0u && dim >= per16) {     // 16 bytes per lane and load
        const uint4* x4 = reinterpret_cast<const uint4*>(x);
        for (uint32_t e = threadIdx.x; e < rows * dim / per16; e += 256) {
            const uint4 q = x4[e];
            const T* qv = reinterpret_cast<const T*>(&q);
#pragma unroll
            for (uint32_t u = 0; u < per16; ++u) {
                const uint32_t g = e * per16 + u, i = g & dmask;
                rec[(g - i) + slot(i)] = qv[u];
            }
        }
    } else {
        for (uint32_t e = threadIdx.x; e < rows * dim; e += 256) {
            const uint32_t i = e & dmask;
            rec[(e - i) + slot(i)] = x[e];
        }
    }
    __syncthreads();
    bool sym = true;
    T* y = out + row0 * dim_f;
    // a lane takes folded columns (the source word and its reverse complement come from the plan's tables: src[dim_f .. 2 dim_f) holds
    // rc_word(src[d]); round 5 - the k-step loop per element and a 32-bit division made this pass vector-ALU bound at k = 6) and walks the
    // workgroup's records: consecutive lanes write consecutive words of a record
    for (uint32_t d = threadIdx.x; d < dim_f; d += 256) {
        const uint32_t w = tsrc[d], wr = tsrc[dim_f + d];
        const uint32_t sw = slot(w), swr = slot(wr);
        for (uint32_t r = 0; r < rows; ++r) {
            T v = (T)0;
            if (w != 0xFFFFFFFFu) {
                v = rec[r * dim + sw];
                sym = sym && (v == rec[r * dim + swr]);
            }
            y[r * dim_f + d] = v;
        }
    }
    if (!sym) atomicOr(asym, PO_FOLD_ASYM);
    // integer counts: can every 128-record block take the equal-total fast path (po_jsd_lut.hip / po_bc_sad.hip)?
    // If so the host need not launch the general tile kernel at all.  Same conditions as the classify kernels.
    if (sizeof(T) == 4 && totals != nullptr) {
        const uint32_t tpr = 256 / rpb;                     // threads per record (rpb is a power of two <= 16)
        const uint32_t r = threadIdx.x / tpr, j = threadIdx.x % tpr;
        if (r < rows) {
            unsigned long long sum = 0;
            uint32_t mx = 0;
            for (uint32_t w = j; w < dim; w += tpr) {
                const uint32_t v = (uint32_t)rec[r * dim + w];
                sum += v;
                mx = max(mx, v);
            }
            // (a record's threads are whole waves or aligned parts of one - tpr is a power of two: partial sums meet inside the wave first,
            //  256 atomics on one LDS word are 256 turns)
            for (uint32_t o = min(tpr, 64u) / 2; o > 0; o >>= 1) {
                sum += __shfl_down(sum, o, 64);
                mx = max(mx, (uint32_t)__shfl_down((int)mx, o, 64));
            }
            if ((j & 63u) == 0u) {
                atomicAdd(&sum_s[r], sum);
                atomicMax(&max_s[r], mx);
            }
        }
        __syncthreads();
        if (j == 0 && r < rows) {
            const uint64_t row = row0 + r;
            // "every tile belongs to the equal-total kernel" needs ONE total for the whole matrix: blocks that are
            // uniform inside but differ from each other still leave their mixed tiles to the general kernel
            const unsigned long long tot = totals[row], ref = totals[0];
            const bool common = tot > 0 && tot == ref;
            uint32_t bits = 0;
            if (!(common && sum_s[r] == tot && max_s[r] <= 255u)) bits |= PO_FOLD_NOT_ALL_TABLE;   // po_jsd_lut.hip: kWideMax
            if (!(common && max_s[r] <= 255u)) bits |= PO_FOLD_NOT_ALL_SAD;
            if (fold_some_equal(totals, row, n, tot)) bits |= PO_FOLD_SOME_EQUAL;
            // (read first: on a uniform assembly EVERY record reports PO_FOLD_SOME_EQUAL, and 50 000 atomics on one address
            // serialise - 65 -> 157 us for this kernel; a stale 0 only costs a redundant atomic)
            if (bits && (*reinterpret_cast<const volatile uint32_t*>(asym) & bits) != bits) atomicOr(asym, bits);
        }
    }
}

// rows too long for LDS (dim > 8192 words): one wave per record, partner reads from global memory
template <typename T>
__global__ __launch_bounds__(256) void rc_fold_long_kernel(const T* __restrict__ in, uint64_t n, uint32_t dim, uint32_t k,
                                                           const uint32_t* __restrict__ src, uint32_t dim_f,
                                                           T* __restrict__ out, uint32_t* __restrict__ asym,
                                                           const unsigned long long* __restrict__ totals) {
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (row >= n) return;
    const T* x = in + row * dim;
    T* y = out + row * dim_f;
    bool sym = true;
    for (uint32_t d = lane; d < dim_f; d += 64) {
        const uint32_t w = src[d];
        T v = (T)0;
        if (w != 0xFFFFFFFFu) {
            v = x[w];
            sym = sym && (v == x[rc_word(w, k)]);
        }
        y[d] = v;
    }
    if (!sym) atomicOr(asym, PO_FOLD_ASYM);
    if (sizeof(T) == 4 && totals != nullptr) {              // see rc_fold_kernel
        unsigned long long sum = 0;
        uint32_t mx = 0;
        for (uint32_t w = lane; w < dim; w += 64) {
            const uint32_t v = (uint32_t)x[w];
            sum += v;
            mx = max(mx, v);
        }
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_down(sum, o, 64);
            mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
        }
        if (lane == 0) {
            // "every tile belongs to the equal-total kernel" needs ONE total for the whole matrix: blocks that are
            // uniform inside but differ from each other still leave their mixed tiles to the general kernel
            const unsigned long long tot = totals[row], ref = totals[0];
            const bool common = tot > 0 && tot == ref;
            uint32_t bits = 0;
            if (!(common && sum == tot && mx <= 255u)) bits |= PO_FOLD_NOT_ALL_TABLE;
            if (!(common && mx <= 255u)) bits |= PO_FOLD_NOT_ALL_SAD;
            if (fold_some_equal(totals, row, n, tot)) bits |= PO_FOLD_SOME_EQUAL;
            // (read first: on a uniform assembly EVERY record reports PO_FOLD_SOME_EQUAL, and 50 000 atomics on one address
            // serialise - 65 -> 157 us for this kernel; a stale 0 only costs a redundant atomic)
            if (bits && (*reinterpret_cast<const volatile uint32_t*>(asym) & bits) != bits) atomicOr(asym, bits);
        }
    }
}

template <typename T>
int launch_fold(po_ctx* ctx, const T* in, uint64_t n, uint32_t dim, uint32_t k, const uint32_t* src, uint32_t dim_f,
                T* out, uint32_t* asym, const unsigned long long* totals) {
    const size_t row_bytes = (size_t)dim * sizeof(T);
    if (row_bytes <= 32768) {
        // records per workgroup: 16 KiB of them (a streaming pass lives on resident workgroups: 32 KiB measured 403 us at k = 6, 16 KiB ...)
        const uint32_t cap = row_bytes <= 16384 ? 16384u : 32768u;
        const uint32_t rpb = (uint32_t)(cap / row_bytes > 16 ? 16 : cap / row_bytes);
        const size_t shmem = rpb * row_bytes + (dim_f <= 1024u ? 2 * (size_t)dim_f * sizeof(uint32_t) : 0);       // records + the plan's two tables
        PO_SHMEM(ctx, rc_fold_kernel<T>, shmem);
        hipLaunchKernelGGL(rc_fold_kernel<T>, dim3((uint32_t)((n + rpb - 1) / rpb)), dim3(256), shmem, ctx->stream,
                           in, n, dim, k, src, dim_f, rpb, out, asym, totals);
    } else {
        hipLaunchKernelGGL(rc_fold_long_kernel<T>, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, ctx->stream, in, n, dim, k,
                           src, dim_f, out, asym, totals);
    }
    PO_CHECK_LAUNCH("rc_fold_kernel");
    return PO_OK;
}

}  // namespace

// k with 4^k == dim, or 0
static uint32_t log4_exact(uint32_t dim) {
    uint32_t k = 0;
    uint64_t d = 1;
    while (d < dim) { d *= 4; ++k; }
    return d == dim ? k : 0;
}

// Builds (or reuses) the source-word table for (dim, gran) and reports the folded width and doubling point.
// gran == PO_FOLD_SELFS_FIRST selects the layout of the Kendall kernel instead: [self-paired words | orbit
// representatives], no padding in between, the row padded to 16 columns at the end (dbl_at = number of self-paired words).
// *dim_f == 0: nothing to fold (dim is not a power of 4).
static int fold_plan(po_ctx* ctx, uint32_t dim, uint32_t gran, uint32_t* dim_f, uint32_t* dbl_at) {
    *dim_f = 0;
    *dbl_at = 0xFFFFFFFFu;
    const uint32_t k = log4_exact(dim);
    if (k == 0 || k > PO_MAX_K) return PO_OK;
    if (ctx->fold_dim == dim && ctx->fold_gran == gran) {
        *dim_f = ctx->fold_dim_f;
        *dbl_at = ctx->fold_dbl_at;
        return PO_OK;
    }
    std::vector<uint32_t> pairs, selfs;
    for (uint32_t w = 0; w < dim; ++w) {
        const uint32_t r = rc_word(w, k);
        if (w < r) pairs.push_back(w);
        else if (w == r) selfs.push_back(w);
    }
    const bool selfs_first = gran == PO_FOLD_SELFS_FIRST;
    const uint32_t ppad = selfs_first ? (uint32_t)selfs.size() : (uint32_t)po_round_up(pairs.size(), gran);
    const uint32_t spad = selfs_first ? (uint32_t)(po_round_up(selfs.size() + pairs.size(), 16) - selfs.size())
                                      : (uint32_t)po_round_up(selfs.size(), 8);
    // A folded matrix that is WIDER than the input is no use - and the workspaces of the JSD / Bray-Curtis path are sized for the
    // input's width: at k = 1, 2 the padding of the two regions (Bray-Curtis: 32 + 8 words for 6 + 4 at k = 2) made the "folded"
    // matrix 2.5 x as wide as the 16 words it came from, and the operands built from it ran past buffers sized for 16 - silently
    // on small inputs, a memory fault at 8 191 records (found by a size sweep in round 5; present since round 1).  Not folded.
    // (The Kendall layout is consumed through its source table only, word pair by word pair: it folds at every k.)
    if (!selfs_first && ppad + spad > dim) {
        ctx->fold_dim = dim;
        ctx->fold_gran = gran;
        ctx->fold_dim_f = 0;
        ctx->fold_dbl_at = 0xFFFFFFFFu;
        return PO_OK;
    }
    std::vector<uint32_t> src(2 * (size_t)(ppad + spad), 0xFFFFFFFFu);     // [source word of every folded column | its reverse complement]
    if (selfs_first) std::swap(pairs, selfs);            // first region: self-paired words, second: representatives
    for (size_t i = 0; i < pairs.size(); ++i) src[i] = pairs[i];
    for (size_t i = 0; i < selfs.size(); ++i) src[ppad + i] = selfs[i];
    for (size_t i = 0; i < (size_t)(ppad + spad); ++i)
        if (src[i] != 0xFFFFFFFFu) src[ppad + spad + i] = rc_word(src[i], k);
    int rc = po_buf_reserve(ctx, &ctx->ws_fold_src, src.size() * sizeof(uint32_t));
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(ctx->ws_fold_src.p, src.data(), src.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));              // src is a local vector
    ctx->fold_dim = dim;
    ctx->fold_gran = gran;
    ctx->fold_dim_f = *dim_f = ppad + spad;
    ctx->fold_dbl_at = *dbl_at = ppad;
    return PO_OK;
}

// number of words that are their own reverse complement (4^(k/2) for even k, 0 for odd k); 0xFFFFFFFF when dim is not 4^k
uint32_t po_fold_selfs(uint32_t dim) {
    const uint32_t k = log4_exact(dim);
    if (k == 0 || k > PO_MAX_K) return 0xFFFFFFFFu;
    return (k & 1u) ? 0u : (1u << k);                      // 4^(k/2) = 2^k
}

// Folds counts (uint32) or frequencies (float64) into ctx->ws_fold if every record is reverse-complement
// symmetric.  On return *folded tells whether ws_fold holds an [n][*dim_f] matrix to use instead of the input.
// Costs one pass over the input and one 4-byte device-to-host read (the only host synchronisation of
// the pairwise entry points; PO_FLAG_NO_RC_FOLD skips it).
// d_totals (may be NULL) and *flags_out (may be NULL): the same pass also tells whether EVERY record has the same
// word total (and small enough counts), i.e. whether every tile qualifies for the equal-total fast paths
// (PO_FOLD_NOT_ALL_TABLE / PO_FOLD_NOT_ALL_SAD clear), so that the caller can leave out the general kernel's launch.  *flags_out == 0xFFFFFFFF: nothing was checked.
int po_rc_fold(po_ctx* ctx, const uint32_t* d_counts, const double* d_freq, const uint64_t* d_totals, uint64_t n,
               uint32_t dim, uint32_t gran, bool* folded, uint32_t* dim_f, uint32_t* dbl_at, uint32_t* flags_out) {
    *folded = false;
    if (flags_out) *flags_out = 0xFFFFFFFFu;
    int rc = fold_plan(ctx, dim, gran, dim_f, dbl_at);
    if (rc || *dim_f == 0 || n == 0) return rc;
    const size_t esz = d_counts ? sizeof(uint32_t) : sizeof(double);
    rc = po_buf_reserve(ctx, &ctx->ws_fold, n * (uint64_t)*dim_f * esz + 256);
    if (rc) return rc;
    if (!ctx->h_flag) PO_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_flag), 64, hipHostMallocDefault));
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_fold.p);
    uint32_t* asym = reinterpret_cast<uint32_t*>(base + po_round_up(n * (uint64_t)*dim_f * esz, 256));
    PO_HIP(hipMemsetAsync(asym, 0, sizeof(uint32_t), ctx->stream));
    const uint32_t k = log4_exact(dim);
    const uint32_t* src = static_cast<const uint32_t*>(ctx->ws_fold_src.p);
    const unsigned long long* tot = reinterpret_cast<const unsigned long long*>(d_totals);
    rc = d_counts ? launch_fold<uint32_t>(ctx, d_counts, n, dim, k, src, *dim_f, reinterpret_cast<uint32_t*>(base), asym, tot)
                  : launch_fold<double>(ctx, d_freq, n, dim, k, src, *dim_f, reinterpret_cast<double*>(base), asym, nullptr);
    if (rc) return rc;
    PO_HIP(hipMemcpyAsync(ctx->h_flag, asym, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    *folded = (*ctx->h_flag & PO_FOLD_ASYM) == 0u;
    if (flags_out && d_counts && d_totals) *flags_out = *ctx->h_flag;
    return PO_OK;
}
