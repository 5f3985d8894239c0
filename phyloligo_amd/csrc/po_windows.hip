// Distance of many profiles (sliding windows) to ONE prototype profile.
//
// Replaces the per-window Python calls of the ContaLocate front end
//   compute_distance_joblib -> JSD / KL / Eucl   (/root/reference/phylopackage/bin/Kount.py:322-330, :69-123)
// (without the x1000 display scaling of :96 and :123, which the host mirror applies).  Work is N_windows x D
// -- five orders of magnitude below the all-by-all matrix -- so this is one wave per window with the
// library float64 log, in the reference's own direct form (term by term, NaN/Inf terms dropped as
// posdef_check_value does, :64-66), reduced over the wave in a fixed order.
#include "po_internal.h"

namespace {

__global__ __launch_bounds__(256) void profile_distance_kernel(const uint32_t* __restrict__ counts,
                                                               const unsigned long long* __restrict__ totals, uint64_t n,
                                                               uint32_t dim, const double* __restrict__ proto, int metric,
                                                               double* __restrict__ out) {
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (w >= n) return;
    const unsigned long long tot = totals[w];
    const uint32_t* row = counts + w * dim;
    double acc = 0.0;
    for (uint32_t d = lane; d < dim; d += 64) {
        const double a = tot ? (double)row[d] / (double)tot : 0.0;     // count2freq: count / kword_count
        const double b = proto[d];
        if (metric == PO_EUCL) {
            const double x = a - b;
            acc += x * x;
        } else if (metric == PO_KL) {                                    // a ln(a/b); NaN / Inf terms are dropped
            if (a > 0.0 && b > 0.0) acc += a * log(a / b);
        } else {                                                         // JSD: 1/2 (KL(a,h) + KL(b,h)), h = (a+b)/2
            const double h = 0.5 * (a + b);
            double t = 0.0;
            if (a > 0.0) t += a * log(a / h);
            if (b > 0.0) t += b * log(b / h);
            acc += t;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) out[w] = (metric == PO_EUCL) ? sqrt(acc) : (metric == PO_JSD ? 0.5 * acc : acc);
}

// out[i] = number of bytes equal to `byte` in [begins[i], ends[i]): the N gate of Kount.py:295 (seq.count("N") / len(seq))
__global__ __launch_bounds__(256) void count_byte_ranges_kernel(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ begins,
                                                                const uint64_t* __restrict__ ends, uint64_t n, uint32_t byte,
                                                                unsigned long long* __restrict__ out) {
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (w >= n) return;
    const uint64_t b = begins[w], e = ends[w];
    uint32_t cnt = 0;
    // unaligned head byte by byte, then 16 bytes per lane and step
    const uint64_t a0 = min(e, (b + 15) & ~(uint64_t)15);
    for (uint64_t p = b + lane; p < a0; p += 64) cnt += seq[p] == byte;
    const uint32_t pat = byte * 0x01010101u;
    uint64_t p = a0 + (uint64_t)lane * 16;
    for (; p + 16 <= e; p += 64 * 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(seq + p);
        const uint32_t x[4] = {v.x ^ pat, v.y ^ pat, v.z ^ pat, v.w ^ pat};
#pragma unroll
        for (int j = 0; j < 4; ++j)                         // zero bytes of x: (x - 0x01..) & ~x & 0x80.. is exact per byte for the count
            cnt += __popc((((x[j] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x[j] | 0x7F7F7F7Fu) ^ 0xFFFFFFFFu);
    }
    const uint64_t tail = a0 + ((e - a0) & ~(uint64_t)15);  // the last partial vector
    for (uint64_t q = tail + lane; q < e; q += 64) cnt += seq[q] == byte;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0) out[w] = cnt;
}

}  // namespace

int po_launch_count_byte_ranges(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_begins, const uint64_t* d_ends, uint64_t n,
                                uint32_t byte, uint64_t* d_out) {
    if (n == 0) return PO_OK;
    hipLaunchKernelGGL(count_byte_ranges_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_seq, d_begins, d_ends, n,
                       byte & 0xFFu, reinterpret_cast<unsigned long long*>(d_out));
    PO_CHECK_LAUNCH("count_byte_ranges_kernel");
    return PO_OK;
}

int po_launch_profile_distances(po_ctx* ctx, const uint32_t* d_counts, const uint64_t* d_totals, uint64_t n, uint32_t dim,
                                const double* d_proto, int metric, double* d_out) {
    if (n == 0) return PO_OK;
    hipLaunchKernelGGL(profile_distance_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_counts,
                       reinterpret_cast<const unsigned long long*>(d_totals), n, dim, d_proto, metric, d_out);
    PO_CHECK_LAUNCH("profile_distance_kernel");
    return PO_OK;
}
