// Frequencies in, integer profiles out - when the frequencies are count2freq output.
//
// The reference's dispatcher hands compute_distances a float64 frequency matrix
// (/root/reference/phylopackage/bin/phyloligo.py:536-553, built by count2freq :633-661 as count / total), so that is
// what a drop-in caller passes to po_pairwise_freq.  Every exact fast path of stage 2 (integer-sum JSD table, packed
// SAD, int8-MFMA Gram, value-histogram ranks) wants the integers back.  For each record this kernel proposes
//     n = rint(1 / smallest positive frequency),   c_w = rint(f_w * n)
// (then n = rint(m / smallest) for m = 2, 3, .. if the smallest count is not 1) - or, round 4, for records whose smallest
// count is large (a 200 kb contig at k = 4: every word occurs hundreds of times, so no m <= 255 exists and the whole matrix
// used to fall back to the float64 kernels: Eucl 12.1 instead of 5.1 ms on the ragged assembly) the denominator that the
// continued fraction of a frequency gives: f = c / n rounded to 53 bits has c / n (in lowest terms) among its convergents
// whenever n^2 < 2^52, the expansion is run exactly on the integers (mantissa, 2^e), one word per lane, and the largest
// denominator found is n / gcd(counts) for all practical purposes - and VERIFIES, bit for bit, that
// (double)c_w / (double)n == f_w  for every word - the very division count2freq and
// prep_transpose_kernel perform.  If every record passes, (c, n) reproduces the caller's matrix exactly and the
// count-based path gives the same distances it would give for those frequencies; if any record fails (frequencies
// from elsewhere, a smallest count above 1 that does not divide the total, NaN, negatives) nothing is assumed
// and the general float64 kernels run.  One pass over the matrix and one flag word read back.
#include "po_internal.h"

namespace {

// Denominator of the first convergent p / q of f (0 < f <= 1) with (double)p / (double)q == f, by the Euclidean algorithm on
// f = M / 2^e exactly (M < 2^53; the first quotient needs 128 bits, everything after it fits 64).  0 when there is none with
// q < 2^40 - the caller then has one candidate less, nothing else.
__device__ uint64_t cf_denominator(double f) {
    if (!(f > 0.0) || f > 1.0) return 0;
    if (f == 1.0) return 1;
    int ex;
    const double mant = frexp(f, &ex);                       // f = mant 2^ex, mant in [0.5, 1), ex <= 0
    if (ex < -60) return 0;
    const uint64_t M = (uint64_t)ldexp(mant, 53);             // exact
    const int e = 53 - ex;                                    // f = M / 2^e, 53 <= e <= 113
    // x = M / A with A = 2^e > M:  a0 = 0;  a1 = floor(A / M), r = A mod M
    const unsigned __int128 A = (unsigned __int128)1 << e;
    const unsigned __int128 q1 = A / M;
    if (q1 >> 40) return 0;
    uint64_t num = M, den = (uint64_t)(A - q1 * M);           // next step divides num by den
    uint64_t p0 = 0, q0 = 1;                                  // convergent i = 0: 0 / 1
    uint64_t p1 = 1, qq1 = (uint64_t)q1;                      // convergent i = 1: 1 / a1
    for (int it = 0; it < 64; ++it) {
        if ((double)p1 / (double)qq1 == f) return qq1;
        if (den == 0) return 0;                               // the expansion ended without reproducing f (cannot happen for q < 2^53)
        const uint64_t a = num / den, r = num - a * den;
        const uint64_t p2 = a * p1 + p0, q2 = a * qq1 + q0;
        if (q2 >> 40) return 0;
        p0 = p1; q0 = qq1; p1 = p2; qq1 = q2;
        num = den; den = r;
    }
    return 0;
}

// one wave per record
__global__ __launch_bounds__(256) void recover_counts_kernel(const double* __restrict__ freq, uint64_t n, uint32_t dim,
                                                             uint32_t* __restrict__ counts,
                                                             unsigned long long* __restrict__ totals,
                                                             uint32_t* __restrict__ bad) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n) return;
    const double* f = freq + r * dim;
    double fmin = 1.0e300;
    bool ok = true;
    for (uint32_t d = lane; d < dim; d += 64) {
        const double v = f[d];
        if (!(v >= 0.0)) ok = false;                         // negative or NaN
        if (v > 0.0 && v < fmin) fmin = v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(fmin, o, 64);        // every lane takes part in the exchange, then chooses
        fmin = fmin < other ? fmin : other;
    }
    // The smallest positive frequency is c_min / n.  Usually c_min = 1; otherwise the candidates c_min = 2, 3, ..
    // are tried in turn (the first that verifies gives the same record in lowest terms or the original one).
    uint32_t* c = counts + r * dim;
    double total = 0.0;                                      // an all-zero record is an empty profile: total 0
    bool found = fmin >= 1.0e300;
    if (found)
        for (uint32_t d = lane; d < dim; d += 64) c[d] = 0u;
    auto verify = [&](double cand) -> bool {                 // (wave uniform) counts = rint(f * cand), every quotient bit for bit
        bool good = true;
        for (uint32_t d = lane; d < dim; d += 64) {
            const double v = f[d];
            const double cd = rint(v * cand);
            if (!(cd < 4294967296.0) || cd / cand != v) good = false;
            c[d] = (cd < 4294967296.0) ? (uint32_t)cd : 0u;
        }
        return __all(good);
    };
    if (!found && ok) {                                      // smallest count 1: the common case for short records
        total = rint(1.0 / fmin);
        if (total >= 1.0 && total < 9.0e15 && 1.0 / total == fmin) found = verify(total);
    }
    if (!found && ok) {
        // The smallest count may be anything: continued fractions.  Every lane expands one positive frequency of the record (its
        // first one); the largest denominator is the total in lowest terms unless every one of those counts shares a factor
        // with it that the others do not - then a small multiple is.
        double mine = 0.0;
        for (uint32_t d = lane; d < dim && !(mine > 0.0); d += 64) mine = f[d];
        uint64_t k = cf_denominator(mine);
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = (uint64_t)__shfl_xor((unsigned long long)k, o, 64);
            k = k > other ? k : other;
        }
        for (uint32_t j = 1; !found && k != 0 && j <= 6u; ++j) {
            total = (double)(k * j);
            found = verify(total);
        }
    }
    for (uint32_t m = 2; !found && ok && m <= 255u; ++m) {   // last resort (totals beyond the reach of the expansion above)
        total = rint((double)m / fmin);
        if (!(total >= 1.0 && total < 9.0e15) || (double)m / total != fmin) continue;
        found = verify(total);
    }
    if (!found) ok = false;
    if (!__all(ok)) {
        if (lane == 0) *bad = 1u;                            // benign race: every writer stores the same value
    } else if (lane == 0) {
        totals[r] = (unsigned long long)total;
    }
}

}  // namespace

// On success *recovered says whether ctx->ws_recover holds counts[n][dim] | totals[n] that reproduce d_freq exactly.
int po_recover_counts(po_ctx* ctx, const double* d_freq, uint64_t n, uint32_t dim, bool* recovered,
                      const uint32_t** d_counts, const uint64_t** d_totals) {
    *recovered = false;
    if (n == 0) return PO_OK;
    const size_t b_counts = po_round_up(n * (uint64_t)dim * sizeof(uint32_t), 256);
    const size_t b_totals = po_round_up(n * sizeof(uint64_t), 256);
    int rc = po_buf_reserve(ctx, &ctx->ws_recover, b_counts + b_totals + 256);
    if (rc) return rc;
    if (!ctx->h_flag) PO_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_flag), 64, hipHostMallocDefault));
    uint8_t* base = static_cast<uint8_t*>(ctx->ws_recover.p);
    uint32_t* counts = reinterpret_cast<uint32_t*>(base);
    unsigned long long* totals = reinterpret_cast<unsigned long long*>(base + b_counts);
    uint32_t* bad = reinterpret_cast<uint32_t*>(base + b_counts + b_totals);
    PO_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(recover_counts_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_freq, n, dim, counts,
                       totals, bad);
    PO_CHECK_LAUNCH("recover_counts_kernel");
    PO_HIP(hipMemcpyAsync(ctx->h_flag + 1, bad, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    *recovered = ctx->h_flag[1] == 0u;
    *d_counts = counts;
    *d_totals = reinterpret_cast<const uint64_t*>(totals);
    return PO_OK;
}
