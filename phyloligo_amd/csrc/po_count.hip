// Stage 1: exact integer spaced-word profiles of every record, straight from the sequence bytes.
//
// Replaces the per-record Python hot loop of the reference
//   select_strand -> upper() -> cut_sequence_and_count_pattern -> count2freq ordering
//   (/root/reference/phylopackage/bin/phyloligo.py:124-149, :683, :601-631, :653)
// with one byte scan: HBM-bound (1 byte per base read once, 4*4^k bytes per record written).
//
// Work decomposition: a record of L bases is cut into ceil(L/2016) chunks, one WAVE per chunk
// (records of a 2 kb assembly are one chunk each; a 10 Mb chromosome is 5k chunks spread over
// the chip).  Fast path (forward words, window <= 16, only A/C/G/T in the chunk; see the kernel): digits by bit
// operations, windows out of a packed register string.  General path: a wave stages its bytes with aligned 16-byte loads, decodes them once to 2-bit
// digits (C=0,G=1,A=2,T=3; anything else breaks a word) in LDS, then every lane slides a 2*W-bit
// rolling register over 32 consecutive window starts.  Waves never wait for each other (no
// workgroup barrier); four of them share a workgroup only to share its launch.  Both strands come out of the same
// pass: the forward register gives the '+' word, a second register filled from the other
// end with complemented digits (digit XOR 1) gives the word the same window spells on the
// reverse-complement strand.  Words go to a private LDS histogram (ds_add_u32) that the wave writes
// out with 16-byte stores; the W-1 words spanning the seq|revcomp(seq) junction of `-s both`
// (phyloligo.py:141) are added by the chunk that holds the record's end.
#include "po_internal.h"

#include <type_traits>

namespace {

constexpr int kPerLane = 32;                       // window starts per lane
constexpr int kTile = 64 * kPerLane;               // 2048 staged positions carrying a window start, per wave
constexpr int kSpan = kTile - 32;                  // window starts per pass: 16 B alignment slack + 16 so that no start of the
                                                   // last lane needs a base beyond the 2048 staged ones when W <= 16 (fast path)
constexpr int kPasses = 1;                         // passes per chunk.  A wave can walk several spans of one record before it
                                                   // flushes its histogram (fewer global atomics on long records); measured at 8:
                                                   // 115 vs 84 us on 2 kb contigs, 397 vs 349 us on a ragged 0.33 Gb assembly -
                                                   // the serial passes expose the load latency that separate waves overlap.
constexpr int kChunkSpan = kSpan * kPasses;
constexpr int kStage = kTile + 64;                 // staged bytes per chunk (halo >= W-1 = 63 behind the last start at 15 + 2015, multiple of 16 B)
constexpr uint32_t kMaxLdsBins = 16384;            // 64 KiB histogram; above that count in HBM directly
// Long records.  The workgroups of a record that crosses workgroups add their histograms into the record's row with global
// atomics; a record of more than ~1 000 chunks (2 Mb) then has hundreds of workgroups on the same 1 KiB of counters and stage 1
// fell from 2.1 TB/s of sequence to 0.87 (2 Mb records), 0.34 (20 Mb), 0.16 (>= 200 Mb; profiles/r04_stage1.txt).  So a record of
// more than kLongChunks chunks adds into SEGMENT rows instead: scratch row s belongs to the record that owns chunk s * kSegChunks,
// a workgroup of that record whose first chunk lies in segment s adds there (at most 32 workgroups per row; the head of the
// record, before its first segment boundary, goes to the record's row as before), and seg_rows_sum_kernel adds 16 segments at a
// time into the record's row - and zeroes them again, so that the scratch is all zeros between calls.
constexpr uint32_t kSegChunks = 128;
constexpr uint32_t kLongChunks = 512;
constexpr uint32_t kSegsPerBlock = 16;
constexpr uint32_t kWaveFillChunks = 256;          // the scan writes no rec_of_chunk entries for records of more chunks: count_kernel searches

struct CountParams {
    uint32_t window, k, dim, nruns;
    uint64_t patbits;      // bit x: window position x is a '1' of the pattern (W <= 64)
    uint32_t sym;          // -s both with a pattern that reads the same in both directions: count the forward words only and
                           // write out hist[w] + hist[rc(w)] (seq + revcomp(seq) is its own reverse complement)
    uint32_t src_shift[PO_MAX_RUNS];
    uint32_t dst_shift[PO_MAX_RUNS];
    uint32_t mask[PO_MAX_RUNS];
    uint32_t le_src[4], le_dst[4];   // the first four runs for a window packed first-base-lowest (fast path)
    uint32_t two_hist;     // -s both with a pattern that is NOT its own mirror image, fast path available: every wave keeps two
                           // histograms, forward words of the pattern and forward words of the REVERSED pattern (= the
                           // minus-strand words relabelled, see the fast path)
    uint32_t marg;         // -s both, spaced pattern of at most 4 positions that is not its own mirror image: the kernel counts the
                           // CONTIGUOUS window (window = k = W, dim = 4^W, symmetric mode: one LDS atomic per start, as cheap as
                           // `1111`) and the write-out sums the 4^(W-k') windows that spell each spaced word (a window counts iff all
                           // its W bases are A/C/G/T, phyloligo.py:601-631, so the spaced profile IS a marginal of the contiguous
                           // one); src_shift / mask / dst_shift / nruns stay those of the spaced pattern, out_dim = 4^k'
    uint32_t out_dim;
    int strand;
    uint32_t n_seqs;
    uint64_t total_bytes;
    uint32_t* seg_rows;              // [segments][row width] scratch rows of long records (all zero between calls), or null
    unsigned long long* seg_tot;     // [segments] their word totals
};

// digit of a base, 4 = not A/C/G/T.  (c>>1)&3 is A0 C1 T2 G3 in either case; 0x72 reorders
// that to C0 G1 A2 T3; membership via a 32-bit set over (upper(c) - 'A').
__device__ __forceinline__ uint32_t base_digit(uint32_t c) {
    const uint32_t x = (c & 0xDFu) - 0x41u;
    const uint32_t member = (x < 32u) ? ((0x00080045u >> x) & 1u) : 0u;   // A=0 C=2 G=6 T=19
    const uint32_t d = (0x72u >> (((c >> 1) & 3u) * 2u)) & 3u;
    return member ? d : 4u;
}

// the byte -> digit map, four entries per dword, for the wave's LDS copy (one 4-byte load per lane instead of four
// evaluations of base_digit: this kernel is bound by instruction issue)
struct DigitTable {
    uint32_t w[64];
    constexpr DigitTable() : w() {
        for (int i = 0; i < 64; ++i) {
            uint32_t e = 0;
            for (int j = 0; j < 4; ++j) {
                const uint32_t c = (uint32_t)(i * 4 + j), x = (c & 0xDFu) - 0x41u;
                const uint32_t member = (x < 32u) ? ((0x00080045u >> x) & 1u) : 0u;
                const uint32_t d = (0x72u >> (((c >> 1) & 3u) * 2u)) & 3u;
                e |= (member ? d : 4u) << (8 * j);
            }
            w[i] = e;
        }
    }
};
__constant__ DigitTable kDigitTable;

// RUNS: -1 = a contiguous k-mer (one run, no gaps): the word is the low 2k bits of the forward register / the
// whole 2W = 2k bit reverse register; 1..4 = that many (shift, mask, shift) runs, unrolled with the run parameters
// in scalar registers; 0 = any number of runs (loop).
template <typename REG, int RUNS>
__device__ __forceinline__ uint32_t word_index(REG reg, const CountParams& P) {
    if (RUNS < 0) return (uint32_t)reg & (P.dim - 1u);
    uint32_t idx = 0;
    if (RUNS > 0) {
#pragma unroll
        for (int r = 0; r < RUNS; ++r) idx |= ((uint32_t)(reg >> P.src_shift[r]) & P.mask[r]) << P.dst_shift[r];
        return idx;
    }
    for (uint32_t r = 0; r < P.nruns; ++r)
        idx |= ((uint32_t)(reg >> P.src_shift[r]) & P.mask[r]) << P.dst_shift[r];
    return idx;
}

// The same for a window packed the other way round (base x of the window at bits [2x, 2x+2), fast path): the result
// is the word index with its k digits in reverse order, digits_reversed() of the index above.
template <int RUNS>
__device__ __forceinline__ uint32_t word_index_le(uint32_t win, const CountParams& P) {
    if (RUNS < 0) return win & (P.dim - 1u);
    uint32_t idx = 0;
#pragma unroll
    for (int r = 0; r < (RUNS > 0 ? RUNS : 1); ++r) idx |= ((win >> P.le_src[r]) & P.mask[r]) << P.le_dst[r];
    return idx;
}
// k base-4 digits of d in reverse order: bit reversal reverses the digits and the two bits inside each; swap those back
__device__ __forceinline__ uint32_t digits_reversed(uint32_t d, uint32_t k) {
    const uint32_t b = __brev(d) >> (32u - 2u * k);
    return ((b & 0x55555555u) << 1) | ((b >> 1) & 0x55555555u);
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_add(uint32_t addr, uint32_t v) {   // ds_add_u32 at a byte address of the LDS
    __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// record i is bytes [begins[i], ends[i]) of the sequence buffer (contiguous records: ends = begins + 1;
// sliding windows: arbitrary, overlapping ranges)
__device__ __forceinline__ uint32_t chunks_of(const uint64_t* begins, const uint64_t* ends, uint32_t i) {
    return (uint32_t)((ends[i] - begins[i] + kChunkSpan - 1) / kChunkSpan);
}

// ---- chunks per record, exclusive scan -> chunk_start[n+1]: three small launches ----------------
// (1) per-1024-record block sums, (2) scan of the block sums, (3) local scan + block offset.
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint64_t* __restrict__ begins,
                                                              const uint64_t* __restrict__ ends, uint32_t n,
                                                              uint32_t* __restrict__ blocksum,
                                                              uint32_t* __restrict__ blockmax) {
    __shared__ uint32_t wsum[4], wmax[4];
    const uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t s = 0, m = 0;
    for (uint32_t e = 0; e < 4; ++e)
        if (base + e < n) { const uint32_t c = chunks_of(begins, ends, base + e); s += c; m = max(m, c); }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); m = max(m, (uint32_t)__shfl_down(m, o, 64)); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wmax[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blocksum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        blockmax[blockIdx.x] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ blocksum, uint32_t nb,
                                                         uint32_t* __restrict__ total,
                                                         const uint32_t* __restrict__ blockmax, uint32_t* __restrict__ max_chunks) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    {                                                               // most chunks in one record (1: no record spans chunks)
        uint32_t m = 0;
        for (uint32_t i = t; i < nb; i += 1024) m = max(m, blockmax[i]);
        part[t] = m;
        __syncthreads();
        for (uint32_t d = 512; d > 0; d >>= 1) {
            if (t < d) part[t] = max(part[t], part[t + d]);
            __syncthreads();
        }
        if (t == 0) *max_chunks = part[0];
        __syncthreads();
    }
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024) {                    // 1024 block sums (1M records) per round
        const uint32_t v = (b0 + t < nb) ? blocksum[b0 + t] : 0u;
        part[t] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t u = (t >= d) ? part[t - d] : 0u;
            __syncthreads();
            part[t] += u;
            __syncthreads();
        }
        if (b0 + t < nb) blocksum[b0 + t] = carry + part[t] - v;    // exclusive
        const uint32_t round_total = part[1023];
        __syncthreads();
        carry += round_total;
    }
    if (t == 0) *total = carry;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const uint64_t* __restrict__ begins,
                                                         const uint64_t* __restrict__ ends, uint32_t n,
                                                         const uint32_t* __restrict__ blocksum,
                                                         const uint32_t* __restrict__ total,
                                                         uint32_t* __restrict__ chunk_start,
                                                         uint32_t* __restrict__ rec_of_chunk) {
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x;
    const uint32_t base = blockIdx.x * 1024 + t * 4;
    uint32_t c[4], s = 0;
    for (uint32_t e = 0; e < 4; ++e) {
        c[e] = (base + e < n) ? chunks_of(begins, ends, base + e) : 0u;
        s += c[e];
    }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t u = (t >= d) ? part[t - d] : 0u;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    uint32_t run = blocksum[blockIdx.x] + part[t] - s;
    // ... and the inverse map chunk -> record, so that a counting wave finds its record with ONE load instead of a 16-step
    // dependent search through chunk_start (round 4: the search was ~2 us of a wave's ~12 us on multi-chunk assemblies).
    // Records of up to 32 chunks are written by their own lane; longer ones (> 64 kb) by the whole wave, one after the other;
    // those of more than kWaveFillChunks (0.5 Mb) not at all (one wave wrote for 150 us on a 1 Gb record, and as long on 500
    // records of 2 Mb that sit in one workgroup of this kernel): count_kernel checks the entry it reads and searches if it is not right.
    uint32_t first[4];
    bool big = false;
    for (uint32_t e = 0; e < 4; ++e) {
        if (base + e < n) chunk_start[base + e] = run;
        first[e] = run;
        if (c[e] <= 32u) { for (uint32_t q = 0; q < c[e]; ++q) rec_of_chunk[run + q] = base + e; }
        else big = true;
        run += c[e];
    }
    unsigned long long todo = __ballot(big);
    const uint32_t lane = t & 63;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t cnt = (uint32_t)__shfl((int)c[e], src, 64), at = (uint32_t)__shfl((int)first[e], src, 64);
            const uint32_t rec = (uint32_t)__shfl((int)(base + e), src, 64);
            if (cnt > 32u && cnt <= kWaveFillChunks) for (uint32_t q = lane; q < cnt; q += 64) rec_of_chunk[at + q] = rec;
        }
    }
    if (blockIdx.x == 0 && t == 0) chunk_start[n] = *total;
}

// ---- the same in ONE launch (round 5) --------------------------------------------------------------------------------------
// Three dependent launches of a few microseconds of work each cost ~15 us of dispatch and drain around a 42 us counting kernel.
// Here a workgroup takes a ticket (its position in the scan; tickets are handed out in the order workgroups START, so a workgroup
// only ever waits for workgroups that are already running - no assumption about how many fit on the chip), publishes the number
// of chunks of its 1024 records and the largest record among them at once, then waits for the aggregates of all tickets before
// its own and adds them up (its lanes poll 256 predecessors at a time).  Aggregates carry the call's epoch in their upper half:
// the state buffer belongs to this kernel alone and entries of earlier calls can never be mistaken for this call's, so nothing
// is cleared between calls; the workgroup with the last ticket - which has seen every aggregate - writes the grand total and the
// largest record and hands the ticket counter back at zero.  The word totals of the records are zeroed on the way (one memset
// less).  Up to kOnePassBlocks workgroups (1 M records: every workgroup reads all aggregates before its own, which is quadratic -
// at 4 096 workgroups of tiny records the three launches above were faster, 3.3 against 4.6 ms); beyond that the three launches.
constexpr uint32_t kOnePassBlocks = 1024;
__global__ __launch_bounds__(256) void scan_chunks_kernel(const uint64_t* __restrict__ begins, const uint64_t* __restrict__ ends, uint32_t n,
                                                          uint32_t nb, unsigned long long* __restrict__ agg, unsigned long long* __restrict__ big,
                                                          uint32_t* __restrict__ ticket, uint32_t* __restrict__ empty_tag, uint32_t epoch,
                                                          uint32_t* __restrict__ chunk_start,
                                                          uint32_t* __restrict__ rec_of_chunk, uint32_t* __restrict__ max_chunks,
                                                          unsigned long long* __restrict__ totals) {
    __shared__ uint32_t part[256], wmax[4], my_ticket, prefix_s, max_s;
    const uint32_t t = threadIdx.x;
    if (t == 0) my_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t blk = my_ticket;
    const uint32_t base = blk * 1024 + t * 4;
    uint32_t c[4], s = 0, m = 0;
    for (uint32_t e = 0; e < 4; ++e) {
        c[e] = (base + e < n) ? chunks_of(begins, ends, base + e) : 0u;
        s += c[e];
        m = max(m, c[e]);
        if (base + e < n) {
            totals[base + e] = 0ull;
            if (c[e] == 0u) *empty_tag = epoch;                     // a record without a chunk: nobody writes its row (zero_rows_kernel)
        }
    }
    part[t] = s;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o, 64));
    if ((t & 63) == 0) wmax[t >> 6] = m;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t u = (t >= d) ? part[t - d] : 0u;
        __syncthreads();
        part[t] += u;
        __syncthreads();
    }
    const unsigned long long tag = (unsigned long long)epoch << 32;
    if (t == 0) {
        const uint32_t bm = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        __hip_atomic_store(&big[blk], tag | bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&agg[blk], tag | part[255], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        prefix_s = 0; max_s = bm;
    }
    __syncthreads();
    // the aggregates of every ticket before this one (each lane polls its own predecessors)
    uint32_t pre = 0, pm = 0;
    for (uint32_t i = t; i < blk; i += 256) {
        unsigned long long v;
        do { v = __hip_atomic_load(&agg[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 32) != epoch);
        pre += (uint32_t)v;
        if (blk == nb - 1) pm = max(pm, (uint32_t)__hip_atomic_load(&big[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    for (int o = 32; o > 0; o >>= 1) { pre += __shfl_down(pre, o, 64); pm = max(pm, (uint32_t)__shfl_down(pm, o, 64)); }
    if ((t & 63) == 0) { atomicAdd(&prefix_s, pre); atomicMax(&max_s, pm); }
    __syncthreads();
    uint32_t run = prefix_s + part[t] - s;
    // the inverse map chunk -> record (see scan_apply_kernel)
    uint32_t first[4];
    bool wide = false;
    for (uint32_t e = 0; e < 4; ++e) {
        if (base + e < n) chunk_start[base + e] = run;
        first[e] = run;
        if (c[e] <= 32u) { for (uint32_t q = 0; q < c[e]; ++q) rec_of_chunk[run + q] = base + e; }
        else wide = true;
        run += c[e];
    }
    unsigned long long todo = __ballot(wide);
    const uint32_t lane = t & 63;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t cnt = (uint32_t)__shfl((int)c[e], src, 64), at = (uint32_t)__shfl((int)first[e], src, 64);
            const uint32_t rec = (uint32_t)__shfl((int)(base + e), src, 64);
            if (cnt > 32u && cnt <= kWaveFillChunks) for (uint32_t q = lane; q < cnt; q += 64) rec_of_chunk[at + q] = rec;
        }
    }
    if (blk == nb - 1 && t == 0) {
        chunk_start[n] = prefix_s + part[255];
        *max_chunks = max_s;
        *ticket = 0u;                                               // every ticket of this call has been taken
    }
}

// Rows that count_kernel ADDS into must be zero first: those of records whose chunks do not all sit in one workgroup of that
// kernel (the others are written whole, with plain stores) and those of records without a chunk (nobody writes them).  With
// word spaces beyond the LDS histogram every row is added into.  A fixed grid walks the matrix 16 bytes per lane; an assembly of
// short contigs (C2: every record one chunk) has nothing to zero and the kernel leaves at its first test - the 51 MB memset of
// rounds 1 - 4 cost 12 us of a 42 us counting kernel there.
__global__ __launch_bounds__(256) void zero_rows_kernel(const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ max_chunks,
                                                        const uint32_t* __restrict__ empty_tag, uint32_t epoch, uint32_t n, uint32_t dim,
                                                        uint32_t wpb, uint32_t* __restrict__ counts) {
    if (*max_chunks <= 1u && *empty_tag != epoch) return;          // every record is one chunk: every row is written whole
    const uint64_t quads_per_row = dim / 4;                        // dim is a power of 4 >= 4
    const uint64_t total = (uint64_t)n * quads_per_row;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (uint64_t)gridDim.x * 256) {
        const uint32_t rec = (uint32_t)(q / quads_per_row);
        const uint32_t f = chunk_start[rec], c = chunk_start[rec + 1] - f;
        const bool whole = *max_chunks > 1u ? (c >= 1u && (f % wpb) + c <= wpb) : c == 1u;
        if (!whole) reinterpret_cast<uint4*>(counts)[q] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// The record of chunk c when some record is longer than kWaveFillChunks chunks and the scan left its entries of rec_of_chunk
// unwritten (whatever the workspace held): an entry is right iff c lies in that record's chunk range; if not, binary search
// through chunk_start (the largest record whose first chunk is <= c; ~20 dependent loads, long records only).
__device__ __forceinline__ uint32_t checked_record_of_chunk(const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ rec_of_chunk,
                                                            uint32_t n_seqs, uint32_t c) {
    uint32_t lo = rec_of_chunk[c];
    bool ok = lo < n_seqs;
    if (ok) ok = chunk_start[lo] <= c && c < chunk_start[lo + 1];
    if (!ok) {
        uint32_t hi = n_seqs;                                       // invariant: chunk_start[lo] <= c < chunk_start[hi]
        lo = 0;
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (chunk_start[mid] <= c) lo = mid; else hi = mid;
        }
    }
    return lo;
}

// ---- counting: one WAVE per chunk, no workgroup barrier -------------------------------------------
// WIDTH: bits of the rolling window registers - 0: 32 (2 W <= 32: every contiguous k-mer up to k = 16), 1: 64 (W <= 32),
//        2: 128 (W <= 64: spaced seeds wider than 32 positions; two 64-bit halves, generic run loop only).
// MODE: which rolling registers the slide keeps - 0 forward only (plus strand, or both strands in symmetric mode),
//       1 reverse only (minus strand), 2 both.
template <bool LDS_HIST, int WIDTH, int MODE, int RUNS>
__global__ __launch_bounds__(1024) void count_kernel(const uint8_t* __restrict__ seq,
                                                         const uint64_t* __restrict__ begins,
                                                         const uint64_t* __restrict__ ends,
                                                         const uint32_t* __restrict__ chunk_start,
                                                         const uint32_t* __restrict__ rec_of_chunk,
                                                         const uint32_t* __restrict__ max_chunks,
                                                         CountParams P, uint32_t waves_per_block,
                                                         uint32_t* __restrict__ counts,
                                                         unsigned long long* __restrict__ totals) {
    extern __shared__ __align__(16) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63;
    // the wave index as a scalar: everything derived from it (chunk, record, lengths, addresses) then lives in scalar
    // registers and is loaded by scalar loads instead of 64 identical vector lanes
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: the histograms of the workgroup's waves first (each aligned to its own size, so that a bin address is an OR),
    // then per wave the staged digits, the byte -> digit table and two slots (junction word, group total)
    constexpr uint32_t kAuxWords = kStage / 4 + 64 + 4;
    const uint32_t HD = ((MODE == 2 && P.two_hist) || P.marg) ? 2u * P.dim : P.dim;      // histogram words per wave
    uint32_t* mine = smem + (LDS_HIST ? waves_per_block * HD : 0) + wave * kAuxWords;
    uint8_t* codes = reinterpret_cast<uint8_t*>(mine);            // [kStage]
    uint8_t* dtab = reinterpret_cast<uint8_t*>(mine + kStage / 4); // [256] byte -> digit

    const uint32_t b = blockIdx.x * waves_per_block + wave;
    const uint32_t nchunks = chunk_start[P.n_seqs];
    // Some record spans several chunks (wave uniform, from the scan): the waves of a workgroup that hold consecutive chunks
    // of ONE record then count into the histogram of the first of them, which writes the record's row once - with plain
    // stores when the whole record lies inside the workgroup (up to 16 chunks = 32 kb), so that global atomics are left to
    // records that cross workgroups.  Two workgroup barriers in that case; none for assemblies of single-chunk records.
    const bool multi = LDS_HIST && *max_chunks > 1u;
    if (b >= nchunks) {                                            // grid is an upper bound
        if (multi) { __syncthreads(); __syncthreads(); }
        return;
    }
    // record of chunk b: the inverse map the scan wrote next to chunk_start (one scalar load)
    const uint32_t lo = *max_chunks > kWaveFillChunks ? checked_record_of_chunk(chunk_start, rec_of_chunk, P.n_seqs, b) : rec_of_chunk[b];
    const uint32_t rec = lo;
    const uint32_t chunk = b - chunk_start[rec];
    const uint32_t rec_chunks = chunk_start[rec + 1] - chunk_start[rec];
    const uint64_t off = begins[rec];
    const int64_t L = (int64_t)(ends[rec] - off);
    const int64_t c_lo = (int64_t)chunk * kChunkSpan;              // window starts [c_lo, c_hi) are ours, kSpan per pass
    const int64_t c_hi = min(c_lo + (int64_t)kChunkSpan, L);
    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)(multi ? wave - min(wave, chunk) : wave));   // first wave of this record in the workgroup
    const bool whole = multi ? (chunk <= wave && wave - chunk + rec_chunks <= waves_per_block) : rec_chunks == 1;
    uint32_t* hist = smem + (LDS_HIST ? lead * HD : 0);            // [dim] (or [2][dim]) when LDS_HIST
    uint32_t* mid_slot = smem + (LDS_HIST ? waves_per_block * HD : 0) + lead * kAuxWords + kStage / 4 + 64;
    // mid_slot[0]: word of the self-mirrored junction window (symmetric mode); mid_slot[1]: words of the group (whole records)

    if (LDS_HIST) {
        uint32_t* own = smem + wave * HD;
        for (uint32_t d = lane * 4; d < HD; d += 256) *reinterpret_cast<uint4*>(own + d) = make_uint4(0, 0, 0, 0);
    }
    if (lane < 2) mine[kStage / 4 + 64 + lane] = lane == 0 ? 0xFFFFFFFFu : 0u;
    if (multi) __syncthreads();

    // ---- stage + decode (aligned 16-byte loads, one table lookup per base) -------------------------
    // the byte -> digit map as a 256-byte LDS table of the wave (each lane evaluates four entries): a lookup costs
    // one LDS read instead of ~8 vector-ALU instructions, and this kernel is ALU-bound
    uint32_t mine_count = 0;                                        // words counted by this lane (general path)
    uint32_t uni_count = 0;                                         // words accounted for by wave-uniform arithmetic (fast path)
    bool slow_junction = false;
    const uint32_t W = P.window;
    constexpr bool want_plus = MODE != 1, want_minus = MODE != 0;
    const uint32_t per_word = P.sym ? 2u : 1u;                      // a forward word also stands for its mirror window
    constexpr bool NARROW = WIDTH == 0;
    constexpr int kMaxW = WIDTH == 2 ? 64 : 32;                     // widest window of this instantiation
    constexpr bool kFastT = LDS_HIST && NARROW && RUNS != 0 && kPasses == 1;
    const bool kFast = kFastT && (MODE != 2 || P.two_hist);
    bool fast_done = false;                                         // wave uniform: this chunk went through the fast path
    // A histogram shared by the chunks of one record is indexed digit-reversed whenever the fast path exists in this
    // kernel (a chunk that has to take the general path then reverses its words one by one); a single-chunk record's
    // histogram is indexed the way its one chunk was counted.
    const bool force_le = kFast && multi && rec_chunks > 1u && (lds_addr(hist) & (P.dim * 4u - 1u)) == 0u;
    // where a minus-strand word m goes in such a histogram: bin (m with every digit complemented) of the reversed-pattern
    // histogram - the bin the fast path's relabelled count of the same window lands in (see the write-out)
    const uint32_t le_cmask = 0x55555555u & (P.dim - 1u), le_rev_base = MODE == 2 ? P.dim : 0u;
    for (int64_t p_lo = c_lo; p_lo < c_hi; p_lo += kSpan) {         // wave uniform; LDS is in order within a wave
    const int64_t p_hi = min(p_lo + (int64_t)kSpan, L);
    const uint64_t a0 = (off + (uint64_t)p_lo) & ~(uint64_t)15;    // 16 B aligned staging origin
    const int64_t pos0 = (int64_t)a0 - (int64_t)off;               // record position of staged byte 0
    __builtin_amdgcn_wave_barrier();
    // ---- fast path: forward words of a window of at most 16 positions, nothing but A/C/G/T in the 2048 staged bytes --
    // Every lane loads its own 32 bases (two aligned 16-byte loads), turns them into digits four at a time with bit
    // operations (no table lookups), packs them into a 64-bit string, takes the next lane's first 16 bases with one DPP
    // move and reads the window of every start out of registers: 3 vector instructions per start (funnel shift, bin
    // address as AND-OR onto the aligned histogram base, the start's bit of the lane's validity mask as the addend).
    // The histogram is indexed by the word with its digits reversed (first base lowest), undone when it is written out.
    // Word totals and the junction words come from wave-uniform arithmetic on the record's last W-1 bases.
    // The minus strand rides on the same register string (SURVEY a-4): the word a window spells on the reverse-complement
    // strand under pattern P is the forward word under the REVERSED pattern with its digits reversed and complemented,
    //     minus(P)[w] == plus(reverse(P))[rc(w)],
    // and the (shift, mask, shift) runs of P for a last-base-lowest window ARE the runs of reverse(P) for a
    // first-base-lowest one: word_index<RUNS>(win) below is that forward word of reverse(P), digit-reversed.  MODE 1 (-s minus)
    // counts those alone, MODE 2 (-s both, pattern not its own mirror image) counts both kinds into two histograms; the
    // write-out relabels: out[d] = H_fwd[digits_reversed(d)] + H_rev[d with every digit complemented].
    const uint32_t hist_addr = lds_addr(hist);
    if (kFast && a0 + kTile <= P.total_bytes && (hist_addr & (P.dim * 4u - 1u)) == 0u) {      // wave uniform
        const uint4 r0 = *reinterpret_cast<const uint4*>(seq + a0 + lane * kPerLane);
        const uint4 r1 = *reinterpret_cast<const uint4*>(seq + a0 + lane * kPerLane + 16);
        const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        uint32_t dg[8], bad = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t code = (w[j] >> 1) & 0x03030303u;                       // A0 C1 T2 G3 in either case
            // bad |= upper(w) ^ the letter that code stands for, as one three-input bit operation (0xF6 = a | (b ^ c))
            bad = __builtin_amdgcn_bitop3_b32(bad, w[j] & 0xDFDFDFDFu, __builtin_amdgcn_perm(0u, 0x47544341u, code), 0xF6);
            dg[j] = __builtin_amdgcn_perm(0u, 0x01030002u, code);                  // C0 G1 A2 T3
        }
        // Round 5: a byte that is not A/C/G/T somewhere in the 2 048 no longer sends the whole chunk to the general path (three
        // times the instructions: 17 % of the chunks of a real assembly - N runs, IUPAC codes - cost as much as the other 83 %).
        // A window counts iff none of its W bases is such a byte; the fast path already adds "the start's bit of the lane's
        // validity mask", so all it takes is clearing the bits of the starts whose window touches one (dirty_starts below) and
        // counting the words that remain instead of computing their number.
        const bool dirty = __any((int)(bad != 0u));
        {
            // 16 digit bytes -> 32 bits, first base lowest
            auto pack16 = [](uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) -> uint32_t {
                const uint32_t t0 = d0 | (d0 >> 6), t1 = d1 | (d1 >> 6), t2 = d2 | (d2 >> 6), t3 = d3 | (d3 >> 6);   // bytes 0, 2: two digits
                const uint32_t e01 = __builtin_amdgcn_perm(t1, t0, 0x06040200u), e23 = __builtin_amdgcn_perm(t3, t2, 0x06040200u);
                const uint32_t f01 = e01 | (e01 >> 4), f23 = e23 | (e23 >> 4);     // bytes 0, 2: four digits
                return __builtin_amdgcn_perm(f23, f01, 0x06040200u);
            };
            const uint32_t plo = pack16(dg[0], dg[1], dg[2], dg[3]), phi = pack16(dg[4], dg[5], dg[6], dg[7]);
            const uint32_t halo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)plo, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
            // starts that count, in staged coordinates: [st_lo, st_hi) (wave uniform), this lane's part as a bit mask
            const int32_t st_lo = (int32_t)(p_lo - pos0);
            const int32_t st_hi = max(st_lo, (int32_t)(min(p_hi, L - (int64_t)W + 1) - pos0));
            const int32_t s_lo = max(st_lo - (int32_t)(lane * kPerLane), 0);
            const int32_t s_hi = min(st_hi - (int32_t)(lane * kPerLane), (int32_t)kPerLane);
            uint32_t vmask = 0;
            if (s_hi > s_lo) {
                const uint32_t span = (uint32_t)(s_hi - s_lo);
                vmask = (span >= 32u ? 0xFFFFFFFFu : ((1u << span) - 1u)) << s_lo;
            }
            uint32_t badcode[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // 0x04 in the byte of every base that is not A/C/G/T (dirty chunks)
            if (!dirty) {
                uni_count += (uint32_t)(st_hi - st_lo) * (MODE == 2 ? 2u : per_word);
            } else {
                uint32_t bm = 0;                                    // bit i: base i of this lane's 32 is not A/C/G/T
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t code = (w[j] >> 1) & 0x03030303u;
                    const uint32_t mism = (w[j] & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x47544341u, code);
                    const uint32_t hb = (((mism & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | mism) & 0x80808080u;   // top bit of every non-zero byte
                    badcode[j] = hb >> 5;
                    bm |= (((hb >> 7) * 0x10204080u) >> 28) << (4 * j);                              // those four bits side by side
                }
                const uint32_t nbm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bm, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
                unsigned long long touched = (unsigned long long)bm | ((unsigned long long)nbm << 32);
                for (uint32_t covered = 1; covered < W;) {           // OR of (touched >> i) for i < W, by doubling (W is wave uniform)
                    const uint32_t step = min(covered, W - covered);
                    touched |= touched >> step;
                    covered += step;
                }
                vmask &= ~(uint32_t)touched;                        // dirty_starts cleared
                uint32_t words = (uint32_t)__popc(vmask);
                for (int o = 32; o > 0; o >>= 1) words += __shfl_down(words, o, 64);
                uni_count += (uint32_t)__builtin_amdgcn_readfirstlane((int)words) * (MODE == 2 ? 2u : per_word);
            }
            if (vmask != 0u) {
                if (RUNS < 0) {
                    const uint32_t m4 = (P.dim - 1u) << 2;
                    uint32_t hbase;                                 // in a vector register: v_and_or_b32 takes one scalar operand
                    asm("v_mov_b32 %0, %1" : "=v"(hbase) : "s"(hist_addr));
#pragma unroll
                    for (int sft = 0; sft < kPerLane; ++sft) {
                        const uint32_t x = sft == 0 ? plo << 2 : sft <= 16 ? __builtin_amdgcn_alignbit(phi, plo, 2 * sft - 2)
                                                                          : __builtin_amdgcn_alignbit(halo, phi, 2 * sft - 34);
                        lds_add((x & m4) | hbase, (vmask >> sft) & 1u);
                    }
                } else {
                    const uint32_t hist2_addr = hist_addr + (MODE == 2 ? P.dim * 4u : 0u);      // reversed-pattern words
#pragma unroll
                    for (int sft = 0; sft < kPerLane; ++sft) {
                        const uint32_t win = sft == 0 ? plo : sft < 16 ? __builtin_amdgcn_alignbit(phi, plo, 2 * sft)
                                           : sft == 16 ? phi : __builtin_amdgcn_alignbit(halo, phi, 2 * sft - 32);
                        if (MODE != 1) lds_add(hist_addr + word_index_le<RUNS>(win, P) * 4u, (vmask >> sft) & 1u);
                        if (MODE != 0) lds_add(hist2_addr + word_index<uint32_t, RUNS>(win, P) * 4u, (vmask >> sft) & 1u);
                    }
                }
            }
            // junction windows of seq + revcomp(seq) (symmetric mode; see the general code below for the pairing rule)
            if (P.strand == PO_STRAND_BOTH && p_hi == L && W > 1) {
                const int64_t ts64 = L - (int64_t)W + 1 - pos0;     // staged index of the first of the record's last W-1 bases
                if (!dirty && ts64 >= 0 && L >= (int64_t)W - 1) { // the W-1 bases exist, are staged (and are bases: a dirty chunk asks the general code)
                    const uint32_t st = (uint32_t)ts64 & 31u, lt = (uint32_t)ts64 >> 5;
                    const uint32_t sel = st < 16u ? __builtin_amdgcn_alignbit(phi, plo, 2u * st)
                                                  : __builtin_amdgcn_alignbit(halo, phi, 2u * st - 32u);
                    const uint32_t tb = 2u * (W - 1u);
                    const uint32_t tmask = (1u << tb) - 1u;
                    const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)sel, (int)lt) & tmask;   // last W-1 bases, first lowest
                    const uint32_t rct = (digits_reversed(tail, W - 1u) ^ 0x55555555u) & tmask;         // their reverse complement
                    const uint64_t J = (uint64_t)tail | ((uint64_t)rct << tb);
                    if (lane < W - 1u) {
                        const uint32_t idx = word_index_le<RUNS>((uint32_t)(J >> (2u * lane)), P);
                        if (MODE == 2) lds_add(hist_addr + idx * 4u, 1u);                  // no mirror pairing: every junction window
                        else if (2u * lane + 2u < W) lds_add(hist_addr + idx * 4u, 1u);
                        else if (2u * lane + 2u == W) *mid_slot = digits_reversed(idx, P.k);
                    }
                    uni_count += MODE == 2 ? W - 1u : 2u * ((W - 1u) >> 1) + ((W & 1u) ? 0u : 1u);
                } else {                                           // shorter record, or the tail starts before the staged range: general code below
                    *reinterpret_cast<uint4*>(codes + lane * kPerLane) = make_uint4(dg[0] | badcode[0], dg[1] | badcode[1], dg[2] | badcode[2], dg[3] | badcode[3]);
                    *reinterpret_cast<uint4*>(codes + lane * kPerLane + 16) = make_uint4(dg[4] | badcode[4], dg[5] | badcode[5], dg[6] | badcode[6], dg[7] | badcode[7]);
                    slow_junction = true;
                }
            }
            fast_done = true;
        }
    }
    if (!fast_done) {
    slow_junction = true;
    reinterpret_cast<uint32_t*>(dtab)[lane] = kDigitTable.w[lane];
    __builtin_amdgcn_wave_barrier();
    // all of the lane's 16-byte loads first (up to three HBM round trips in flight at once), then the decoding
    constexpr int kVecPerLane = (kStage / 16 + 63) / 64;
    uint4 raw[kVecPerLane];
#pragma unroll
    for (int it = 0; it < kVecPerLane; ++it) {
        const uint32_t v = lane + 64 * it;
        const uint64_t a = a0 + (uint64_t)v * 16;
        raw[it] = make_uint4(0, 0, 0, 0);
        if (v < kStage / 16) {
            if (a + 16 <= P.total_bytes) {
                raw[it] = *reinterpret_cast<const uint4*>(seq + a);
            } else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int i = 0; i < 16; ++i)
                    if (a + i < P.total_bytes) w[i >> 2] |= (uint32_t)seq[a + i] << (8 * (i & 3));
                raw[it] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    uint32_t sep_seen = 0;                                          // any byte that is not A/C/G/T among the staged ones
#pragma unroll
    for (int it = 0; it < kVecPerLane; ++it) {
        const uint32_t v = lane + 64 * it;
        if (v >= kStage / 16) break;
        const uint32_t w[4] = {raw[it].x, raw[it].y, raw[it].z, raw[it].w};
        const int64_t left = L - (pos0 + (int64_t)v * 16);          // bytes of this vector that belong to the record
        const int32_t nvalid = (int32_t)max((int64_t)0, min(left, (int64_t)16));
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (uint32_t)dtab[w[j] & 0xFFu] | ((uint32_t)dtab[(w[j] >> 8) & 0xFFu] << 8) |
                   ((uint32_t)dtab[(w[j] >> 16) & 0xFFu] << 16) | ((uint32_t)dtab[w[j] >> 24] << 24);
        sep_seen |= (o[0] | o[1] | o[2] | o[3]) & 0x04040404u;
        if (nvalid < 16) {                                          // the vector holding the record end (and beyond): separators
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (j * 4 + i >= nvalid) o[j] = (o[j] & ~(0xFFu << (8 * i))) | (4u << (8 * i));
        }
        *reinterpret_cast<uint4*>(codes + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __builtin_amdgcn_wave_barrier();                               // LDS is in order within a wave

    // ---- slide: lane owns starts [32 lane, 32 lane + 32) ---------------------------------------------
    {
        typedef typename std::conditional<WIDTH == 0, uint32_t, typename std::conditional<WIDTH == 1, uint64_t, unsigned __int128>::type>::type reg_t;
        // (windows wider than 32 positions - rare spaced seeds - walk their digits out of LDS in a rolled loop further down:
        // 95 unrolled steps on two 64-bit halves would triple the code of this translation unit for nothing)
        constexpr int kCwVecs = WIDTH == 2 ? 0 : (kPerLane + kMaxW - 1 + 15) / 16;      // 16-byte vectors holding the lane's 32 starts + W - 1 more digits
        uint32_t cw[4 * kCwVecs + 1];
#pragma unroll
        for (int q = 0; q < kCwVecs; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(codes + lane * kPerLane + 16 * q);
            cw[4 * q] = v.x; cw[4 * q + 1] = v.y; cw[4 * q + 2] = v.z; cw[4 * q + 3] = v.w;
        }
        // this lane's window starts s = 0..31 sit at record positions first + s; those in [p_lo, p_hi) are ours
        const int64_t first = pos0 + (int64_t)lane * kPerLane;
        const int64_t lo64 = p_lo - first, hi64 = p_hi - first;
        const int32_t s_lo = (int32_t)max(lo64, (int64_t)0), s_hi = (int32_t)min(hi64, (int64_t)kPerLane);
        reg_t fwd = 0, rev = 0;
        const uint32_t top = 2 * W - 2;
        // No byte other than A/C/G/T anywhere in the staged range (the usual case; decided per wave): every window that
        // lies inside the record is a word, so the per-position run length and its tests go away - a start s counts iff
        // s_lo <= s < s_hi' with s_hi' cut at the last start whose window ends inside the record.  Half the vector
        // instructions of the general loop below (this kernel is bound by instruction issue, not by bytes).
        if constexpr (WIDTH == 2) {
            uint32_t run = 0;
            const uint8_t* mycodes = codes + lane * kPerLane;
            for (uint32_t i = 0; i < kPerLane + W - 1; ++i) {
                const uint32_t d = mycodes[i];
                run = (d < 4u) ? run + 1u : 0u;
                if (want_plus) fwd = (fwd << 2) | (reg_t)(d & 3u);
                if (want_minus) rev = (rev >> 2) | ((reg_t)((d & 3u) ^ 1u) << top);
                const int s = (int)i - (int)(W - 1);
                if (s >= s_lo && s < s_hi && run >= W) {
                    if (want_plus) {
                        const uint32_t idx = word_index<reg_t, RUNS>(fwd, P);
                        if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                        else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                        mine_count += per_word;
                    }
                    if (want_minus) {
                        const uint32_t idx = word_index<reg_t, RUNS>(rev, P);
                        if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                        else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                        ++mine_count;
                    }
                }
            }
        } else {
        if (!__any((int)(sep_seen != 0u))) {
            const int64_t hi_valid = min(p_hi, L - (int64_t)W + 1) - first;
            const int32_t s_hi2 = (int32_t)max((int64_t)s_lo, min(hi_valid, (int64_t)kPerLane));
            const uint32_t span = (uint32_t)(s_hi2 - s_lo);              // starts of this lane that count
            mine_count += span * ((want_plus ? per_word : 0u) + (want_minus ? 1u : 0u));
#pragma unroll
            for (int i = 0; i < kPerLane + kMaxW - 1; ++i) {
                if (i < (int)(kPerLane + W - 1)) {                      // uniform
                    const uint32_t d = (cw[i >> 2] >> (8 * (i & 3))) & 3u;
                    if (want_plus) fwd = (fwd << 2) | (reg_t)d;
                    if (want_minus) rev = (rev >> 2) | ((reg_t)(d ^ 1u) << top);
                    const int s = i - (int)(W - 1);                     // window start index of this lane (uniform)
                    if (s >= 0 && (uint32_t)(s - s_lo) < span) {
                        if (want_plus) {
                            uint32_t idx = word_index<reg_t, RUNS>(fwd, P);
                            if (force_le) idx = digits_reversed(idx, P.k);
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                        }
                        if (want_minus) {
                            uint32_t idx = word_index<reg_t, RUNS>(rev, P);
                            if (force_le) idx = (idx ^ le_cmask) + le_rev_base;
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                        }
                    }
                }
            }
        } else {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < kPerLane + kMaxW - 1; ++i) {
            if (i < (int)(kPerLane + W - 1)) {                      // uniform
                const uint32_t d = (cw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                run = (d < 4u) ? run + 1u : 0u;
                if (want_plus) fwd = (fwd << 2) | (reg_t)(d & 3u);
                if (want_minus) rev = (rev >> 2) | ((reg_t)((d & 3u) ^ 1u) << top);
                const int s = i - (int)(W - 1);                     // window start index of this lane (uniform)
                if (s >= 0) {
                    if (run >= W && s >= s_lo && s < s_hi) {
                        if (want_plus) {
                            uint32_t idx = word_index<reg_t, RUNS>(fwd, P);
                            if (force_le) idx = digits_reversed(idx, P.k);
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                            mine_count += per_word;
                        }
                        if (want_minus) {
                            uint32_t idx = word_index<reg_t, RUNS>(rev, P);
                            if (force_le) idx = (idx ^ le_cmask) + le_rev_base;
                            if (LDS_HIST) atomicAdd(&hist[idx], 1u);
                            else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                            ++mine_count;
                        }
                    }
                }
            }
        }
        }
        }   // WIDTH != 2
    }

    }   // general path
    __builtin_amdgcn_wave_barrier();

    // ---- junction words of seq + revcomp(seq) (-s both), by the chunk holding the record end ----
    // Symmetric mode: the W-1 junction windows are mirror images of each other (start p <-> 2L - W - p), so only the
    // first of each pair is added (the write-out doubles it); for even W the middle window is its own mirror and
    // spells a self-paired word, which is added once, after the doubling (mid_word).
    if (slow_junction && P.strand == PO_STRAND_BOTH && p_hi == L && lane < W - 1) {    // the pass that holds the record end
        const int64_t p = L - (int64_t)W + 1 + (int64_t)lane;       // start in the 2L-long virtual string
        const bool first_of_pair = 2 * p < 2 * L - (int64_t)W, middle = 2 * p == 2 * L - (int64_t)W;
        if (p >= 0 && p < L && p + (int64_t)W <= 2 * L && (!P.sym || first_of_pair || middle)) {
            uint32_t idx = 0;
            bool ok = true;
            for (uint32_t x = 0; x < W; ++x) {
                const int64_t q = p + x;
                const bool fw = q < L;
                const int64_t qpos = fw ? q : 2 * L - 1 - q;        // record position of the base (the last W-1 of the record)
                uint32_t d = (qpos >= pos0) ? (uint32_t)codes[qpos - pos0]               // staged digits of this chunk
                                            : base_digit(seq[off + (uint64_t)qpos]);     // short last chunk: before the staged range
                ok = ok && (d < 4u);
                if (!fw) d ^= 1u;
                if ((P.patbits >> x) & 1ull) idx = idx * 4u + (d & 3u);
            }
            if (ok) {
                if (P.sym && middle) {
                    *mid_slot = idx;
                    ++mine_count;
                } else {
                    if (LDS_HIST) atomicAdd(&hist[(fast_done || force_le) ? digits_reversed(idx, P.k) : idx], 1u);
                    else atomicAdd(&counts[(uint64_t)rec * P.dim + idx], 1u);
                    mine_count += per_word;
                }
            }
        }
    }

    }   // passes

    // ---- totals: wave reduce, one store / atomic per chunk ----------------------------------------
    if (slow_junction)                                              // wave uniform: somebody counted per lane
        for (int o = 32; o > 0; o >>= 1) mine_count += __shfl_down(mine_count, o, 64);
    mine_count += uni_count;
    if (lane == 0) {
        if (rec_chunks == 1) totals[rec] = mine_count;
        else if (whole) atomicAdd(&mid_slot[1], mine_count);       // LDS: the group's first wave writes the sum below
        else if (mine_count) {
            unsigned long long* tot = &totals[rec];
            if (P.seg_rows != nullptr && rec_chunks > kLongChunks && (b / kSegChunks) * kSegChunks >= chunk_start[rec])
                tot = &P.seg_tot[b / kSegChunks];                   // long record: its segment's total (seg_rows_sum_kernel)
            atomicAdd(tot, (unsigned long long)mine_count);
        }
    }

    // ---- flush: the first wave of the record's group, once every wave of the group has counted ------
    if (multi) __syncthreads();
    if (LDS_HIST && wave == lead) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t width = P.marg ? P.out_dim : P.dim;
        uint32_t* row = counts + (uint64_t)rec * width;
        if (P.seg_rows != nullptr && rec_chunks > kLongChunks && (b / kSegChunks) * kSegChunks >= chunk_start[rec])
            row = P.seg_rows + (uint64_t)(b / kSegChunks) * width;  // long record (never `whole`: atomics below)
        const uint32_t mid_word = P.sym ? mid_slot[0] : 0xFFFFFFFFu;
        if (whole && rec_chunks > 1 && lane == 0) totals[rec] = mid_slot[1];
        const bool le = fast_done || force_le;
        auto bin = [&](uint32_t d) -> uint32_t {
            if (!P.sym) return hist[d];
            // reverse complement in the C,G,A,T digit coding: digits in reverse order, each XOR 1.  Bit reversal reverses
            // the digits and the two bits inside each; swapping those back and flipping the low bit of every digit is
            // one swap of odd and even bits with the even result bits complemented.
            const uint32_t b = __brev(d) >> (32u - 2u * P.k);
            const uint32_t r = (((b & 0x55555555u) << 1) | (((b >> 1) & 0x55555555u) ^ 0x55555555u)) & (P.dim - 1u);
            return hist[d] + hist[r] + (d == mid_word ? 1u : 0u);
        };
        if (P.marg) {
            // contiguous windows -> spaced words: every window x (both strands: hist[x] + hist[rc(x)], + 1 for the junction's
            // self-mirrored window) is added to the spaced word it spells, in a second LDS array behind the histogram
            uint32_t* outh = hist + P.dim;                        // [out_dim], zeroed with the histogram
            const uint32_t cmaskw = 0x55555555u & (P.dim - 1u);
            for (uint32_t h = lane; h < P.dim; h += 64) {
                const uint32_t d = le ? digits_reversed(h, P.k) : h;                       // the window, first base highest
                // slot of its reverse complement rc(d) = digits reversed, each complemented: rc(d) itself, or digits_reversed(rc(d)) = d ^ c
                const uint32_t hr = le ? (d ^ cmaskw) : (digits_reversed(d, P.k) ^ cmaskw);
                const uint32_t v = hist[h] + hist[hr] + (d == mid_word ? 1u : 0u);
                uint32_t word = 0;
                for (uint32_t r = 0; r < P.nruns; ++r) word |= ((d >> P.src_shift[r]) & P.mask[r]) << P.dst_shift[r];
                if (v) atomicAdd(&outh[word], v);
            }
            __builtin_amdgcn_wave_barrier();
            if (whole) {
                for (uint32_t d0 = lane * 4; d0 < P.out_dim; d0 += 256) {
                    if (d0 + 4 <= P.out_dim) *reinterpret_cast<uint4*>(row + d0) = *reinterpret_cast<const uint4*>(outh + d0);
                    else for (uint32_t d = d0; d < P.out_dim; ++d) row[d] = outh[d];
                }
            } else {
                for (uint32_t d = lane; d < P.out_dim; d += 64) {
                    const uint32_t v = outh[d];
                    if (v) atomicAdd(&row[d], v);
                }
            }
        } else
        if (le) {
            // the histogram is indexed by the digit-reversed word: outputs d0..d0+3 differ in their last digit = the first
            // of the reversed word; the mirror window's word, reversed, is d with every digit complemented: one aligned quad.
            // The same quad of the reversed-pattern histogram holds the minus-strand counts of d0..d0+3 (MODE 1, 2).
            const uint32_t cmask = 0x55555555u & (P.dim - 1u), top = 2u * P.k - 2u;
            const uint32_t* hrev = hist + (MODE == 2 ? P.dim : 0u);
            if (whole) {
                for (uint32_t d0 = lane * 4; d0 < P.dim; d0 += 256) {
                    const uint32_t base = digits_reversed(d0, P.k);
                    uint32_t v[4] = {0, 0, 0, 0};
                    if (MODE != 1) {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) v[j] = hist[base + (j << top)];
                    }
                    if (P.sym || MODE != 0) {
                        const uint4 q = *reinterpret_cast<const uint4*>(hrev + ((d0 ^ cmask) & ~3u));
                        v[0] += q.y; v[1] += q.x; v[2] += q.w; v[3] += q.z;
                    }
                    if (P.sym) {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) v[j] += (d0 + j == mid_word) ? 1u : 0u;
                    }
                    *reinterpret_cast<uint4*>(row + d0) = make_uint4(v[0], v[1], v[2], v[3]);
                }
            } else {                                               // consecutive lanes, consecutive bins: dense atomics
                for (uint32_t d = lane; d < P.dim; d += 64) {
                    uint32_t v = MODE != 1 ? hist[digits_reversed(d, P.k)] : 0u;
                    if (P.sym || MODE != 0) v += hrev[d ^ cmask];
                    if (P.sym) v += (d == mid_word ? 1u : 0u);
                    if (v) atomicAdd(&row[d], v);
                }
            }
        } else
        if (whole) {
            for (uint32_t d = lane * 4; d < P.dim; d += 256)
                *reinterpret_cast<uint4*>(row + d) = make_uint4(bin(d), bin(d + 1), bin(d + 2), bin(d + 3));
        } else {
            for (uint32_t d = lane; d < P.dim; d += 64) {
                const uint32_t v = bin(d);
                if (v) atomicAdd(&row[d], v);
            }
        }
    }
}

// Long records (see kSegChunks): the segment rows of 16 consecutive segments are added into the rows of the records that own
// them - one atomic per bin and run of segments with one owner - and zeroed again.  A segment whose first chunk belongs to a
// record of at most kLongChunks chunks was never written and is not read.
__global__ __launch_bounds__(256) void seg_rows_sum_kernel(const uint32_t* __restrict__ chunk_start, const uint32_t* __restrict__ rec_of_chunk,
                                                           const uint32_t* __restrict__ max_chunks, uint32_t n_seqs, uint32_t width,
                                                           uint32_t* __restrict__ seg_rows, unsigned long long* __restrict__ seg_tot,
                                                           uint32_t* __restrict__ counts, unsigned long long* __restrict__ totals) {
    if (*max_chunks <= kLongChunks) return;                        // no long record in this input (uniform)
    __shared__ uint32_t owner[kSegsPerBlock];
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t nchunks = chunk_start[n_seqs];
    const uint32_t s0 = blockIdx.x * kSegsPerBlock, t = threadIdx.x;
    if (t < kSegsPerBlock) {
        const uint64_t c = (uint64_t)(s0 + t) * kSegChunks;
        uint32_t o = NONE;
        if (c < nchunks) {
            const uint32_t r = checked_record_of_chunk(chunk_start, rec_of_chunk, n_seqs, (uint32_t)c);
            if (chunk_start[r + 1] - chunk_start[r] > kLongChunks) o = r;
        }
        owner[t] = o;
    }
    __syncthreads();
    bool any = false;
    for (uint32_t s = 0; s < kSegsPerBlock; ++s) any = any || owner[s] != NONE;
    if (!any) return;                                              // uniform
    if (t == 0) {
        uint32_t cur = NONE;
        unsigned long long acc = 0;
        for (uint32_t s = 0; s < kSegsPerBlock; ++s) {
            const uint32_t o = owner[s];
            if (o != cur) {
                if (cur != NONE && acc) atomicAdd(&totals[cur], acc);
                cur = o; acc = 0;
            }
            if (o != NONE) { acc += seg_tot[s0 + s]; seg_tot[s0 + s] = 0; }
        }
        if (cur != NONE && acc) atomicAdd(&totals[cur], acc);
    }
    for (uint32_t d = t; d < width; d += 256) {
        uint32_t v[kSegsPerBlock];
#pragma unroll
        for (uint32_t s = 0; s < kSegsPerBlock; ++s)               // independent loads first
            v[s] = owner[s] != NONE ? seg_rows[(uint64_t)(s0 + s) * width + d] : 0u;
        uint32_t cur = NONE, acc = 0;
#pragma unroll
        for (uint32_t s = 0; s < kSegsPerBlock; ++s) {
            const uint32_t o = owner[s];
            if (o != cur) {
                if (cur != NONE && acc) atomicAdd(&counts[(uint64_t)cur * width + d], acc);
                cur = o; acc = 0;
            }
            if (v[s]) { acc += v[s]; seg_rows[(uint64_t)(s0 + s) * width + d] = 0u; }
        }
        if (cur != NONE && acc) atomicAdd(&counts[(uint64_t)cur * width + d], acc);
    }
}

}  // namespace

int po_launch_count(po_ctx* ctx, const uint8_t* d_seq, const uint64_t* d_begins, const uint64_t* d_ends, uint64_t n_seqs,
                    uint64_t total_bytes, uint64_t sum_lengths, const po_pattern& pat, int strand, uint32_t* d_counts,
                    uint64_t* d_totals) {
    if (n_seqs == 0) return PO_OK;
    if (n_seqs >= (1ull << 31)) { po_set_error("too many records (%llu)", (unsigned long long)n_seqs); return PO_EUNSUPPORTED; }
    const uint64_t max_chunks = sum_lengths / kChunkSpan + n_seqs;
    if (max_chunks >= (1ull << 31)) { po_set_error("input too large for one launch"); return PO_EUNSUPPORTED; }

    const uint32_t nb = (uint32_t)((n_seqs + 1023) / 1024);
    int rc = po_buf_reserve(ctx, &ctx->ws_aux, (n_seqs + 1 + 2 * (uint64_t)nb + 2 + max_chunks) * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t* chunk_start = static_cast<uint32_t*>(ctx->ws_aux.p);
    uint32_t* blocksum = chunk_start + n_seqs + 1;
    uint32_t* total = blocksum + nb;
    uint32_t* blockmax = total + 1;
    uint32_t* d_max_chunks = blockmax + nb;
    uint32_t* rec_of_chunk = d_max_chunks + 1;                     // [max_chunks]: chunk -> record, written by scan_apply_kernel

    const bool one_pass = nb <= kOnePassBlocks && pat.dim >= 4 && pat.dim <= kMaxLdsBins;
    const bool zero_by_kernel = one_pass && (reinterpret_cast<uintptr_t>(d_counts) & 15u) == 0;      // (its stores are 16 bytes wide)
    if (one_pass && !zero_by_kernel) PO_HIP(hipMemsetAsync(d_counts, 0, n_seqs * (uint64_t)pat.dim * sizeof(uint32_t), ctx->stream));
    if (one_pass) {
        // one launch for the scan (which also zeroes the word totals); the rows that need zeros get them further down, once the
        // launch geometry of the counting kernel is known
        const size_t need = (2 * (size_t)kOnePassBlocks) * sizeof(unsigned long long) + 64;
        if (ctx->ws_scan.cap < need) {
            rc = po_buf_reserve(ctx, &ctx->ws_scan, need);
            if (rc) return rc;
            PO_HIP(hipMemsetAsync(ctx->ws_scan.p, 0, ctx->ws_scan.cap, ctx->stream));
            ctx->scan_epoch = 0;
        }
        if (++ctx->scan_epoch == 0u) {                              // the tag wrapped: start over from a clean buffer
            PO_HIP(hipMemsetAsync(ctx->ws_scan.p, 0, ctx->ws_scan.cap, ctx->stream));
            ctx->scan_epoch = 1;
        }
        unsigned long long* agg = static_cast<unsigned long long*>(ctx->ws_scan.p);
        unsigned long long* big = agg + kOnePassBlocks;
        uint32_t* ticket = reinterpret_cast<uint32_t*>(big + kOnePassBlocks);
        hipLaunchKernelGGL(scan_chunks_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_begins, d_ends, (uint32_t)n_seqs, nb, agg, big, ticket,
                           ticket + 1, ctx->scan_epoch, chunk_start, rec_of_chunk, d_max_chunks, reinterpret_cast<unsigned long long*>(d_totals));
        PO_CHECK_LAUNCH("scan_chunks_kernel");
    } else {
        // every row and every total zeroed, three launches for the scan
        PO_HIP(hipMemsetAsync(d_counts, 0, n_seqs * (uint64_t)pat.dim * sizeof(uint32_t), ctx->stream));
        PO_HIP(hipMemsetAsync(d_totals, 0, n_seqs * sizeof(uint64_t), ctx->stream));
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_begins, d_ends, (uint32_t)n_seqs, blocksum, blockmax);
        PO_CHECK_LAUNCH("scan_block_sums_kernel");
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, blocksum, nb, total, blockmax, d_max_chunks);
        PO_CHECK_LAUNCH("scan_sums_kernel");
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_begins, d_ends, (uint32_t)n_seqs, blocksum, total, chunk_start, rec_of_chunk);
        PO_CHECK_LAUNCH("scan_apply_kernel");
    }

    CountParams P;
    memset(&P, 0, sizeof(P));
    P.window = pat.window; P.k = pat.k; P.dim = pat.dim; P.nruns = pat.nruns;
    for (uint32_t r = 0; r < pat.nruns; ++r) {
        P.src_shift[r] = pat.src_shift[r]; P.dst_shift[r] = pat.dst_shift[r]; P.mask[r] = pat.mask[r];
    }
    for (uint32_t r = 0; r < pat.nruns && r < 4; ++r) {            // the same runs for a window packed first-base-lowest
        const uint32_t len = (uint32_t)__builtin_popcount(pat.mask[r]) / 2;
        P.le_src[r] = 2 * (pat.window - len) - pat.src_shift[r];    // 2 x (first window position of the run)
        P.le_dst[r] = 2 * (pat.k - len) - pat.dst_shift[r];         // 2 x (rank of its first digit)
    }
    for (uint32_t i = 0; i < pat.k; ++i) P.patbits |= 1ull << pat.ones[i];
    P.strand = strand;
    bool palindromic = true;
    for (uint32_t x = 0; x < pat.window; ++x) palindromic = palindromic && (((P.patbits >> x) & 1ull) == ((P.patbits >> (pat.window - 1 - x)) & 1ull));
    P.sym = (strand == PO_STRAND_BOTH && palindromic && pat.dim <= kMaxLdsBins) ? 1u : 0u;
    P.n_seqs = (uint32_t)n_seqs;
    P.total_bytes = total_bytes;

    const bool lds_hist = pat.dim <= kMaxLdsBins;
    // segment rows of long records: only when the input can hold one; zeroed when (re)allocated, kept zero by seg_rows_sum_kernel
    const uint64_t n_segs = max_chunks / kSegChunks + 1;
    const bool seg = lds_hist && max_chunks > kLongChunks;
    if (seg) {
        const size_t cap_before = ctx->ws_seg.cap;
        rc = po_buf_reserve(ctx, &ctx->ws_seg, n_segs * ((uint64_t)pat.dim * sizeof(uint32_t) + sizeof(uint64_t)));
        if (rc) return rc;
        if (ctx->ws_seg.cap != cap_before) PO_HIP(hipMemsetAsync(ctx->ws_seg.p, 0, ctx->ws_seg.cap, ctx->stream));
        P.seg_tot = static_cast<unsigned long long*>(ctx->ws_seg.p);                 // 8-byte entries first
        P.seg_rows = reinterpret_cast<uint32_t*>(P.seg_tot + n_segs);
    }
    // -s both, spaced pattern of at most 4 positions, not its own mirror image (1101, 1011): count contiguous windows,
    // marginalise at the write-out - `1101 both` 79.2 -> 50.0 us at C2 (two histograms with two atomics per start before).
    // Five positions measured WORSE than two histograms (10011 both 145 us, 11101 both 133 us against 68 us for the contiguous
    // 11111: 1 024 windows per record to fold, 16 per lane): the write-out has to stay small against the 2 016 starts of a chunk.
    if (strand == PO_STRAND_BOTH && !palindromic && pat.window <= 4 && pat.window > pat.k) {
        P.marg = 1u;
        P.out_dim = pat.dim;
        P.k = pat.window;
        P.dim = 1u << (2 * pat.window);
        P.patbits = (1ull << pat.window) - 1ull;
        P.sym = 1u;
    }
    // -s both, pattern not its own mirror image, fast path available (window <= 16, at most 4 runs): two histograms per wave
    P.two_hist = (strand == PO_STRAND_BOTH && !P.sym && lds_hist && 2 * pat.window <= 32 && pat.nruns <= 4 && pat.dim <= 4096) ? 1u : 0u;
    const size_t per_wave = kStage + 256 + 16 + (lds_hist ? (size_t)P.dim * 4 * ((P.two_hist || P.marg) ? 2 : 1) : 0);
    uint32_t wpb = (uint32_t)((80u << 10) / per_wave);                // waves per workgroup within 80 KiB of LDS
    wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);                          // up to 4 consecutive chunks of one record share a flush (8 and 16 measured slower)
    const size_t shmem = per_wave * wpb;
    const uint32_t grid = (uint32_t)((max_chunks + wpb - 1) / wpb);
    unsigned long long* tot = reinterpret_cast<unsigned long long*>(d_totals);
    if (zero_by_kernel) {
        const uint32_t* empty_tag = reinterpret_cast<const uint32_t*>(static_cast<unsigned long long*>(ctx->ws_scan.p) + 2 * kOnePassBlocks) + 1;
        hipLaunchKernelGGL(zero_rows_kernel, dim3(2048), dim3(256), 0, ctx->stream, chunk_start, d_max_chunks, empty_tag, ctx->scan_epoch,
                           (uint32_t)n_seqs, pat.dim, wpb, d_counts);
        PO_CHECK_LAUNCH("zero_rows_kernel");
    }
    const int width = 2 * pat.window <= 32 ? 0 : (pat.window <= 32 ? 1 : 2);
    const int mode = (strand == PO_STRAND_PLUS || P.sym) ? 0 : (strand == PO_STRAND_MINUS ? 1 : 2);
    auto launch = [&](auto k) -> int {
        PO_SHMEM(ctx, k, shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64 * wpb), shmem, ctx->stream, d_seq, d_begins, d_ends, chunk_start, rec_of_chunk, d_max_chunks, P, wpb, d_counts, tot);
        return PO_OK;
    };
    int lrc = PO_OK;
    // contiguous k-mer: one run that takes the low 2k bits as they are (the reverse register then holds exactly 2W = 2k bits)
    const bool simple = pat.nruns == 1 && pat.src_shift[0] == 0 && pat.dst_shift[0] == 0 && pat.window == pat.k;
    const int runs = width == 2 ? 0 : ((simple || P.marg) ? -1 : (pat.nruns <= 4 ? (int)pat.nruns : 0));
#define PO_COUNT_RUNS(L, N, M, R) if (runs == R) lrc = launch(count_kernel<L, N, M, R>);
#define PO_COUNT_CASE(L, N, M)                                                                                  \
    if (lds_hist == L && width == N && mode == M) {                                                             \
        PO_COUNT_RUNS(L, N, M, -1) PO_COUNT_RUNS(L, N, M, 0) PO_COUNT_RUNS(L, N, M, 1) PO_COUNT_RUNS(L, N, M, 2) \
        PO_COUNT_RUNS(L, N, M, 3) PO_COUNT_RUNS(L, N, M, 4)                                                      \
    }
#define PO_COUNT_WIDE(L, M) if (lds_hist == L && width == 2 && mode == M) lrc = launch(count_kernel<L, 2, M, 0>);
    PO_COUNT_CASE(true, 0, 0) PO_COUNT_CASE(true, 0, 1) PO_COUNT_CASE(true, 0, 2)
    PO_COUNT_CASE(true, 1, 0) PO_COUNT_CASE(true, 1, 1) PO_COUNT_CASE(true, 1, 2)
    PO_COUNT_CASE(false, 0, 0) PO_COUNT_CASE(false, 0, 1) PO_COUNT_CASE(false, 0, 2)
    PO_COUNT_CASE(false, 1, 0) PO_COUNT_CASE(false, 1, 1) PO_COUNT_CASE(false, 1, 2)
    PO_COUNT_WIDE(true, 0) PO_COUNT_WIDE(true, 1) PO_COUNT_WIDE(true, 2)
    PO_COUNT_WIDE(false, 0) PO_COUNT_WIDE(false, 1) PO_COUNT_WIDE(false, 2)
#undef PO_COUNT_RUNS
#undef PO_COUNT_CASE
#undef PO_COUNT_WIDE
    if (lrc) return lrc;
    PO_CHECK_LAUNCH("count_kernel");
    if (seg) {
        hipLaunchKernelGGL(seg_rows_sum_kernel, dim3((uint32_t)((n_segs + kSegsPerBlock - 1) / kSegsPerBlock)), dim3(256), 0, ctx->stream,
                           chunk_start, rec_of_chunk, d_max_chunks, (uint32_t)n_seqs, pat.dim, P.seg_rows, P.seg_tot, d_counts, tot);
        PO_CHECK_LAUNCH("seg_rows_sum_kernel");
    }
    return PO_OK;
}
