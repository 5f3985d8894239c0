// Stage 2, rank metrics: Kendall tau (KT) tiles, and the per-record order statistics both
// rank metrics need (average ranks for Spearman, tie counts for Kendall).
//
// Replaces  phylodist.KT  (/root/reference/phylopackage/core/phylodist.py:71-74):
//     1 - Bio.Cluster.distancematrix((a,b), dist="k")[1][0]
// The C Clustering Library's Kendall distance is 1 - tau with
//     tau = (con - dis) / sqrt((con+dis+exx) (con+dis+exy)),   distance 1 if a factor is 0,
// con/dis = concordant/discordant word pairs, exx/exy = pairs tied in one vector only.  With
// T = D(D-1)/2 word pairs and t_x = pairs tied in x:  con+dis+exx = T - t_y, con+dis+exy = T - t_x,
// so only  S = con - dis = sum_{p<q} sgn(x_p - x_q) sgn(y_p - y_q)  depends on the record pair.
// Ranks and ties of frequencies equal those of the integer counts (same positive divisor per
// record), so everything here is exact integer work until the final division.
#include "po_tiles.h"

namespace {

constexpr int kThreads = 256;
constexpr int PT = 16;            // 16 x 16 record pairs per workgroup, one pair per lane
constexpr int CH = 128;           // words per staged chunk
constexpr int kRow = CH + 1;      // +1: rows of a chunk start in different banks

constexpr uint32_t kValueBins = 8192;   // counts below this take the histogram path of row_order_kernel (64 KiB of LDS)

// One workgroup per record, input row major (uint32 counts or float64 frequencies).
//   rt[d][npad]      centred average rank of word d:  #less + (#equal - D)/2
//   lessrank[r][d]   #less (an integer with the order and ties of the input)
//   r2[r][d]         2 #less + #equal - D = twice the centred average rank, an exact integer (po_gram_i8.hip)
//   rowstat[3][r]    number of word pairs tied in the record
// Integer counts below kValueBins (decided per record, on the device): #less and #equal come from the
// histogram of the record's count VALUES and its prefix sum - O(D + max count) instead of the O(D^2)
// all-pairs comparison that float64 input (and very long records) still use.
template <typename T, bool COUNTS>
__global__ __launch_bounds__(kThreads) void row_order_kernel(const T* __restrict__ rows, uint64_t n, uint32_t dim,
                                                             uint64_t npad, double* __restrict__ rt,
                                                             uint32_t* __restrict__ lessrank, int32_t* __restrict__ r2,
                                                             double* __restrict__ rowstat) {
    extern __shared__ __align__(16) unsigned char dyn[];                // COUNTS: hist[kValueBins] | pref[kValueBins]
    __shared__ T chunk[2048];
    __shared__ unsigned long long tied;
    __shared__ uint32_t red[kThreads];
    const uint64_t r = blockIdx.x;
    const uint32_t t = threadIdx.x;
    if (t == 0) tied = 0;
    const T* row = rows + r * dim;
    unsigned long long my_tied = 0;
    auto emit = [&](uint32_t d, uint32_t less, uint32_t equal) {
        if (rt) rt[(uint64_t)d * npad + r] = (double)less + 0.5 * ((double)equal - (double)dim);
        if (lessrank) lessrank[r * dim + d] = less;
        if (r2) r2[r * dim + d] = (int32_t)(2u * less + equal) - (int32_t)dim;
        my_tied += equal - 1;                                        // ordered tied partners of word d
    };

    bool by_histogram = false;
    if (COUNTS) {
        uint32_t* hist = reinterpret_cast<uint32_t*>(dyn);
        uint32_t* pref = hist + kValueBins;
        uint32_t mx = 0;
        for (uint32_t d = t; d < dim; d += kThreads) mx = max(mx, (uint32_t)row[d]);
        red[t] = mx;
        __syncthreads();
        for (uint32_t s = kThreads / 2; s > 0; s >>= 1) {
            if (t < s) red[t] = max(red[t], red[t + s]);
            __syncthreads();
        }
        const uint32_t vmax = red[0];                                // uniform over the workgroup
        __syncthreads();
        by_histogram = vmax < kValueBins;
        if (by_histogram) {
            const uint32_t nv = vmax + 1;
            for (uint32_t v = t; v < nv; v += kThreads) hist[v] = 0;
            __syncthreads();
            for (uint32_t d = t; d < dim; d += kThreads) atomicAdd(&hist[(uint32_t)row[d]], 1u);
            __syncthreads();
            // exclusive prefix sum of hist[0..nv): contiguous segment per lane, scan of the 256 segment sums
            const uint32_t seg = (nv + kThreads - 1) / kThreads;
            const uint32_t v0 = min(t * seg, nv), v1 = min(v0 + seg, nv);
            uint32_t ssum = 0;
            for (uint32_t v = v0; v < v1; ++v) ssum += hist[v];
            red[t] = ssum;
            __syncthreads();
            for (uint32_t dlt = 1; dlt < kThreads; dlt <<= 1) {
                const uint32_t u = (t >= dlt) ? red[t - dlt] : 0u;
                __syncthreads();
                red[t] += u;
                __syncthreads();
            }
            uint32_t run = red[t] - ssum;
            for (uint32_t v = v0; v < v1; ++v) { pref[v] = run; run += hist[v]; }
            __syncthreads();
            for (uint32_t d = t; d < dim; d += kThreads) {
                const uint32_t v = (uint32_t)row[d];
                emit(d, pref[v], hist[v]);
            }
        }
    }
    if (!by_histogram) {
        for (uint32_t base = 0; base < dim; base += kThreads) {          // words owned by lanes this round
            const uint32_t d = base + t;
            const T x = (d < dim) ? row[d] : T(0);
            uint32_t less = 0, equal = 0;
            for (uint32_t c0 = 0; c0 < dim; c0 += 2048) {
                const uint32_t len = min(2048u, dim - c0);
                __syncthreads();
                for (uint32_t i = t; i < len; i += kThreads) chunk[i] = row[c0 + i];
                __syncthreads();
                for (uint32_t i = 0; i < len; ++i) {
                    const T y = chunk[i];
                    less += (y < x);
                    equal += (y == x);
                }
            }
            if (d < dim) emit(d, less, equal);
        }
    }
    for (int o = 32; o > 0; o >>= 1) my_tied += __shfl_down(my_tied, o, 64);
    __syncthreads();
    if ((t & 63u) == 0) atomicAdd(&tied, my_tied);
    __syncthreads();
    if (t == 0) rowstat[3 * npad + r] = (double)(tied / 2);
}

// The same for integer counts of up to 1 024 words (k <= 5) with ONE WAVE per record, four records per workgroup and no workgroup barrier
// (end of round 5): at D = 256 the kernel above spends a 256-lane workgroup, 64 KiB of LDS and sixteen barriers on a record - 348 us for
// 50 000 records, 0.35 of the 2.5 ms a float32 Spearman matrix takes.  A lane holds the words lane, lane + 64, ... in registers; counts
// below 1 024 go through the wave's own histogram (4 KiB) and its prefix sum (a scan over the lanes), larger ones (contigs beyond ~40 kb
// at k = 4) through the all-pairs comparison inside the wave.  Same integers out.
constexpr uint32_t kWaveBins = 1024, kWaveDim = 1024;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kThreads) void row_order_wave_kernel(const uint32_t* __restrict__ rows, uint64_t n, uint32_t dim,
                                                                  uint64_t npad, double* __restrict__ rt,
                                                                  uint32_t* __restrict__ lessrank, int32_t* __restrict__ r2,
                                                                  double* __restrict__ rowstat) {
    __shared__ uint32_t lds[kThreads / 64][2 * kWaveBins];              // per wave: hist | pref   (all-pairs path: the record's values)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * (kThreads / 64) + wave;
    if (r >= n) return;                                                 // (no workgroup barrier below)
    uint32_t* hist = lds[wave];
    uint32_t* pref = hist + kWaveBins;
    const uint32_t* row = rows + r * dim;
    constexpr int W = kWaveDim / 64;
    uint32_t val[W];
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t d = lane + 64u * i;
        val[i] = d < dim ? row[d] : 0u;
        mx = max(mx, val[i]);
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    unsigned long long my_tied = 0;
    auto emit = [&](uint32_t d, uint32_t less, uint32_t equal) {
        if (rt) rt[(uint64_t)d * npad + r] = (double)less + 0.5 * ((double)equal - (double)dim);
        if (lessrank) lessrank[r * dim + d] = less;
        if (r2) r2[r * dim + d] = (int32_t)(2u * less + equal) - (int32_t)dim;
        my_tied += equal - 1;
    };
    if (mx < kWaveBins) {
        const uint32_t nv = mx + 1;
        for (uint32_t v = lane; v < nv; v += 64) hist[v] = 0;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (lane + 64u * i < dim) atomicAdd(&hist[val[i]], 1u);
        wave_lds_sync();
        const uint32_t seg = (nv + 63u) / 64u;
        const uint32_t v0 = min(lane * seg, nv), v1 = min(v0 + seg, nv);
        uint32_t ssum = 0;
        for (uint32_t v = v0; v < v1; ++v) ssum += hist[v];
        uint32_t incl = ssum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)lane >= o) incl += u;
        }
        uint32_t run = incl - ssum;
        for (uint32_t v = v0; v < v1; ++v) { pref[v] = run; run += hist[v]; }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint32_t d = lane + 64u * i;
            if (d < dim) emit(d, pref[val[i]], hist[val[i]]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (lane + 64u * i < dim) hist[lane + 64u * i] = val[i];
        wave_lds_sync();
#pragma unroll 1
        for (int i = 0; i < W; ++i) {
            const uint32_t d = lane + 64u * i;
            if (d >= dim) break;
            const uint32_t x = val[i];
            uint32_t less = 0, equal = 0;
            for (uint32_t q = 0; q < dim; ++q) {
                const uint32_t y = hist[q];
                less += (y < x);
                equal += (y == x);
            }
            emit(d, less, equal);
        }
    }
    for (int o = 32; o > 0; o >>= 1) my_tied += __shfl_down(my_tied, o, 64);
    if (lane == 0) rowstat[3 * npad + r] = (double)(my_tied / 2);
}

__global__ __launch_bounds__(kThreads) void zero_pad_kernel(double* __restrict__ rt, uint64_t n, uint32_t dim,
                                                            uint64_t npad) {
    const uint64_t pad = npad - n;
    const uint64_t total = pad * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
        rt[(i / pad) * npad + n + (i % pad)] = 0.0;
}


__device__ __forceinline__ int sgn_i32(int v) { return min(max(v, -1), 1); }   // v_med3_i32

template <typename OUT>
__global__ __launch_bounds__(kThreads) void kt_tile_kernel(const uint32_t* __restrict__ counts, po_tile_args A) {
    __shared__ int xa_p[PT][kRow], xb_p[PT][kRow], xa_q[PT][kRow], xb_q[PT][kRow];
    const uint32_t t = threadIdx.x;
    const uint32_t pi = t >> 4, pj = t & 15;
    uint32_t ti, tj;
    po_tile_coords(A, PT, blockIdx.x, ti, tj);
    const uint64_t i0 = (uint64_t)ti * PT, j0 = (uint64_t)tj * PT;

    auto load_chunk = [&](int (*dst)[kRow], uint64_t r0, uint32_t c0) {
        for (uint32_t e = t; e < PT * CH; e += kThreads) {
            const uint32_t rr = e / CH, cc = e % CH;
            const uint64_t r = r0 + rr;
            dst[rr][cc] = (r < A.n && c0 + cc < A.dim) ? (int)counts[r * A.dim + c0 + cc] : 0;
        }
    };

    long long S = 0;
    for (uint32_t p0 = 0; p0 < A.dim; p0 += CH) {
        __syncthreads();
        load_chunk(xa_p, i0, p0);
        load_chunk(xb_p, j0, p0);
        const uint32_t plen = min((uint32_t)CH, A.dim - p0);
        for (uint32_t q0 = p0; q0 < A.dim; q0 += CH) {
            __syncthreads();
            load_chunk(xa_q, i0, q0);
            load_chunk(xb_q, j0, q0);
            __syncthreads();
            const uint32_t qlen = min((uint32_t)CH, A.dim - q0);
            int s = 0;
            for (uint32_t p = 0; p < plen; ++p) {
                const int xp = xa_p[pi][p], yp = xb_p[pj][p];
                const uint32_t qb = (q0 == p0) ? p + 1 : 0;
                for (uint32_t q = qb; q < qlen; ++q)
                    s += sgn_i32(xa_q[pi][q] - xp) * sgn_i32(xb_q[pj][q] - yp);
            }
            S += s;
        }
    }

    const uint64_t i = i0 + pi, j = j0 + pj;
    if (i >= A.n || j >= A.n || !po_in_block(A, i, j)) return;
    const double T = 0.5 * (double)A.dim * ((double)A.dim - 1.0);
    const double* ties = A.rowstat + 3 * A.npad;
    const double dx = T - ties[j], dy = T - ties[i];     // con+dis+exx, con+dis+exy
    const double v = po_kt_value((double)S, dy, dx, po_kt_rs(dy), po_kt_rs(dx));
    po_store_pair<OUT>(A, i, j, v, po_tile_mirrors(A, ti, tj));
}

}  // namespace

int po_launch_ranks(po_ctx* ctx, const uint32_t* d_counts, const double* d_freq, uint64_t n, uint32_t dim,
                    uint64_t npad, double* d_rt, uint32_t* d_lessrank, int32_t* d_r2, double* d_rowstat) {
    if (n == 0) return PO_OK;
    if (d_rt && (dim & 7u))   // operand rows are padded to a multiple of 8 words
        PO_HIP(hipMemsetAsync(d_rt + (uint64_t)dim * npad, 0, (po_round_up(dim, 8) - dim) * npad * sizeof(double), ctx->stream));
    if (d_rt && npad > n) {
        hipLaunchKernelGGL(zero_pad_kernel, dim3(256), dim3(kThreads), 0, ctx->stream, d_rt, n, dim, npad);
        PO_CHECK_LAUNCH("zero_pad_kernel");
    }
    if (d_counts && dim <= kWaveDim) {
        hipLaunchKernelGGL(row_order_wave_kernel, dim3((uint32_t)((n + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, ctx->stream,
                           d_counts, n, dim, npad, d_rt, d_lessrank, d_r2, d_rowstat);
    } else if (d_counts) {
        auto k = row_order_kernel<uint32_t, true>;
        const size_t shmem = 2 * kValueBins * sizeof(uint32_t);
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)n), dim3(kThreads), shmem, ctx->stream, d_counts, n, dim, npad, d_rt, d_lessrank,
                           d_r2, d_rowstat);
    } else {
        hipLaunchKernelGGL((row_order_kernel<double, false>), dim3((uint32_t)n), dim3(kThreads), 0, ctx->stream, d_freq, n, dim,
                           npad, d_rt, d_lessrank, d_r2, d_rowstat);
    }
    PO_CHECK_LAUNCH("row_order_kernel");
    return PO_OK;
}

int po_launch_kt(po_ctx* ctx, const uint32_t* d_lessrank, uint64_t n, uint32_t dim, const po_tile_args& a,
                 uint64_t* tiles) {
    (void)n; (void)dim;
    const uint64_t nblocks = po_tile_count(a, PT);
    if (tiles) *tiles += nblocks;
    if (nblocks == 0) return PO_OK;
    if (nblocks >= (1ull << 31)) { po_set_error("too many tiles for one launch (%llu)", (unsigned long long)nblocks); return PO_EUNSUPPORTED; }
    if (a.out_f32)
        hipLaunchKernelGGL(kt_tile_kernel<float>, dim3((uint32_t)nblocks), dim3(kThreads), 0, ctx->stream, d_lessrank, a);
    else
        hipLaunchKernelGGL(kt_tile_kernel<double>, dim3((uint32_t)nblocks), dim3(kThreads), 0, ctx->stream, d_lessrank, a);
    PO_CHECK_LAUNCH("kt_tile_kernel");
    return PO_OK;
}
