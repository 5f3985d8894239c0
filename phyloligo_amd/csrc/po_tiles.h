// Tile bookkeeping shared by the stage-2 kernels: which tile a workgroup owns inside a block of the
// matrix, and where a computed pair is stored (and mirrored).
#pragma once

#include "po_internal.h"

// host: number of workgroups for a block with square tiles of `edge` records
static inline uint64_t po_tile_count(const po_tile_args& a, uint32_t edge) {
    const uint64_t r0 = a.row_begin / edge, r1 = (a.row_end + edge - 1) / edge;
    const uint64_t c0 = a.col_begin / edge, c1 = (a.col_end + edge - 1) / edge;
    if (a.row_end <= a.row_begin || a.col_end <= a.col_begin) return 0;
    if (a.triangular) return (r1 - r0) * (r1 - r0 + 1) / 2;
    return (r1 - r0) * (c1 - c0);
}

#if defined(__HIPCC__)
// ---- LDS helpers -----------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) unsigned char po_lds_byte;
typedef const __attribute__((address_space(1))) unsigned char po_glb_byte;
// raw LDS byte address of a __shared__ pointer
__device__ __forceinline__ uint32_t po_lds_addr(const void* p) { return (uint32_t)(uintptr_t)((const po_lds_byte*)p); }
__device__ __forceinline__ double2 po_lds_read_d2(uint32_t addr) {
    return *((const __attribute__((address_space(3))) double2*)(uintptr_t)addr);
}
__device__ __forceinline__ double po_lds_read_f64(uint32_t addr) {
    return *((const __attribute__((address_space(3))) double*)(uintptr_t)addr);
}
// LDS-DMA: 64 lanes x 16 B straight from global memory into 1 KiB of LDS at `lds_base` (wave uniform)
__device__ __forceinline__ void po_glds16(const void* gptr, void* lds_base) {
    __builtin_amdgcn_global_load_lds((po_glb_byte*)gptr, (po_lds_byte*)lds_base, 16, 0, 0);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which on gfx9
// (loads and stores share the in-order vmcnt counter) makes every wave wait for all of its outstanding GLOBAL
// stores as well - in an epilogue that alternates LDS transposes with output stores that is one full store
// latency per barrier.  Not for loops that stage with LDS-DMA (those complete on vmcnt).
__device__ __forceinline__ void po_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ __forceinline__ void po_lds_barrier() {}
__device__ __forceinline__ uint32_t po_lds_addr(const void*) { return 0; }
__device__ __forceinline__ double2 po_lds_read_d2(uint32_t) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ double po_lds_read_f64(uint32_t) { return 0.0; }
__device__ __forceinline__ void po_glds16(const void*, void*) {}
#endif

// sqrt of a non-negative, normal-range float64: v_rsq_f64 seed (2^-23 accurate) + one coupled
// Goldschmidt step + one residual correction -> within 1 ulp; exact 0 for 0 (no denormal scaling needed:
// squared distances of frequency vectors are 0 or >= 1e-16).
__device__ __forceinline__ double po_sqrt_nonneg(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// ---- Kendall's tau from S = con - dis and the per-record factors (shared by every Kendall kernel so that they
// agree bit for bit).  d = T - t = the number of word pairs NOT tied in the record (an integer);
//     tau = S / sqrt(d_r d_c) = S * (rs_r * rs_c),   rs = 1 / sqrt(d)  (0 for d = 0: the reference returns
// distance 1, i.e. KT 0, when a factor vanishes).  The product of the two per-record terms is formed first, so
// the value does not depend on which record is the row; S == d_r == d_c (same order, same ties) is exactly 1.
// KT = 1 - (1 - tau) as in phylodist.KT (1 - distance).  An f64 sqrt and a divide per matrix entry (about 50
// instructions) measured as much vector time as the whole matrix-core Gram of po_pairdot.hip; this is 6.
__device__ __forceinline__ double po_kt_rs(double d) {            // 1/sqrt(d): v_rsq_f64 seed + two Newton steps
    double y = __builtin_amdgcn_rsq(d);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    return d > 0.0 ? y : 0.0;
}
__device__ __forceinline__ double po_kt_value(double S, double dr, double dc, double rsr, double rsc) {
    double tau = S * (rsr * rsc);
    tau = fmin(fmax(tau, -1.0), 1.0);
    if (dr == dc && dr > 0.0) {
        if (S == dr) tau = 1.0;
        if (-S == dr) tau = -1.0;
    }
    return 1.0 - (1.0 - tau);
}

// linear id -> tile of the upper triangle of a T x T tile grid (tj >= ti), row major
__device__ __forceinline__ void po_tri_decode(uint64_t b, uint32_t T, uint32_t& ti, uint32_t& tj) {
    const double tt = 2.0 * T + 1.0;
    uint32_t i = (uint32_t)((tt - sqrt(tt * tt - 8.0 * (double)b)) * 0.5);
    auto before = [T](uint64_t r) { return r * T - r * (r - 1) / 2; };   // tiles in rows < r
    while (i > 0 && before(i) > b) --i;
    while (before((uint64_t)i + 1) <= b) ++i;
    ti = i;
    tj = (uint32_t)(i + (b - before(i)));
}

// ---- workgroup -> tile order ---------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs, each with a private 4 MiB L2 (observed, not a
// contract: this only affects speed).  po_xcd_swizzle gives every XCD one contiguous range of the
// logical tile order, and the logical order walks the block in bands of kBand tile rows, column by
// column inside a band, so the tiles an XCD works on at any moment share their row operands (the band)
// and, eight at a time, their column operand.
constexpr uint32_t kXcds = 8;
constexpr uint32_t kBand = 8;

__device__ __forceinline__ uint64_t po_xcd_swizzle(uint64_t b, uint64_t nb) {
    const uint64_t x = b % kXcds, q = b / kXcds;
    const uint64_t base = nb / kXcds, rem = nb % kXcds;
    return x * base + (x < rem ? x : rem) + q;                        // bijective for any nb
}

// logical index -> (ti, tj) inside a T x T upper triangle (tj >= ti), banded order
__device__ __forceinline__ void po_tri_band_decode(uint64_t L, uint32_t T, uint32_t& ti, uint32_t& tj) {
    const uint64_t S = kBand;
    // tiles in bands < r:  r*S(S+1)/2 + S*(r*T - S*r(r+1)/2)   (bands of S full rows; the last may be short)
    auto before = [T](uint64_t r) { const uint64_t S = kBand; return r * (S * (S + 1) / 2) + S * (r * T - S * r * (r + 1) / 2); };
    const uint64_t nbands = (T + S - 1) / S;
    // estimate by solving the quadratic, then fix up
    const double a = 0.5 * (double)(S * S), bq = (double)S * T + 0.5 * (double)S - 0.5 * (double)(S * S);
    double disc = bq * bq - 4.0 * a * (double)L;
    uint64_t r = (uint64_t)((bq - sqrt(disc > 0.0 ? disc : 0.0)) / (2.0 * a));
    if (r >= nbands) r = nbands - 1;
    while (r > 0 && before(r) > L) --r;
    while (r + 1 < nbands && before(r + 1) <= L) ++r;
    uint64_t l = L - before(r);
    const uint64_t r0 = r * S;
    const uint64_t Se = (T - r0 < S) ? (T - r0) : S;                   // rows in this band
    const uint64_t tri = Se * (Se + 1) / 2;
    if (l < tri) {                                                     // the band's own triangle: column d has d+1 tiles
        uint64_t d = (uint64_t)((sqrt(8.0 * (double)l + 1.0) - 1.0) * 0.5);
        while (d * (d + 1) / 2 > l) --d;
        while ((d + 1) * (d + 2) / 2 <= l) ++d;
        ti = (uint32_t)(r0 + (l - d * (d + 1) / 2));
        tj = (uint32_t)(r0 + d);
    } else {
        l -= tri;
        ti = (uint32_t)(r0 + l % Se);
        tj = (uint32_t)(r0 + Se + l / Se);
    }
}

// absolute tile indices of logical position L in the block's tile order (see po_xcd_swizzle)
__device__ __forceinline__ void po_tile_coords_logical(const po_tile_args& A, uint32_t edge, uint64_t L, uint32_t& ti, uint32_t& tj) {
    const uint32_t r0 = (uint32_t)(A.row_begin / edge), r1 = (uint32_t)((A.row_end + edge - 1) / edge);
    const uint32_t c0 = (uint32_t)(A.col_begin / edge), c1 = (uint32_t)((A.col_end + edge - 1) / edge);
    if (A.triangular) {
        const uint32_t T = r1 - r0;
        po_tri_band_decode(L, T, ti, tj);
        ti += r0;
        tj += r0;
    } else {
        const uint64_t tr = r1 - r0, tc = c1 - c0;
        const uint64_t band = L / (kBand * tc), l = L % (kBand * tc);
        const uint64_t Se = (tr - band * kBand < kBand) ? (tr - band * kBand) : kBand;
        ti = r0 + (uint32_t)(band * kBand + l % Se);
        tj = c0 + (uint32_t)(l / Se);
    }
}

// absolute tile indices of workgroup `b`
__device__ __forceinline__ void po_tile_coords(const po_tile_args& A, uint32_t edge, uint64_t b, uint32_t& ti, uint32_t& tj) {
    const uint32_t r0 = (uint32_t)(A.row_begin / edge), r1 = (uint32_t)((A.row_end + edge - 1) / edge);
    const uint32_t c0 = (uint32_t)(A.col_begin / edge), c1 = (uint32_t)((A.col_end + edge - 1) / edge);
    const uint64_t total = A.triangular ? (uint64_t)(r1 - r0) * (r1 - r0 + 1) / 2 : (uint64_t)(r1 - r0) * (c1 - c0);
    po_tile_coords_logical(A, edge, po_xcd_swizzle(b, total), ti, tj);
}

// Output stores are non-temporal: the matrix is written once and never read by the kernel that writes it, and 20 GB of
// it streaming through the L2s / Infinity Cache with the default policy evict the operands the tiles keep re-reading
// (measured on the pair-dot Kendall kernel: 8.6 -> 7.7 ms).
#if defined(PO_NO_NT_STORES)
template <typename T> __device__ __forceinline__ void po_out_store(T* p, T v) { *p = v; }
__device__ __forceinline__ void po_store2(double* p, double a, double b) { *reinterpret_cast<double2*>(p) = make_double2(a, b); }
__device__ __forceinline__ void po_store2(float* p, double a, double b) { *reinterpret_cast<float2*>(p) = make_float2((float)a, (float)b); }
#else
typedef double po_d2v __attribute__((ext_vector_type(2)));
typedef float po_f2v __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ void po_out_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void po_store2(double* p, double a, double b) { const po_d2v v = {a, b}; __builtin_nontemporal_store(v, reinterpret_cast<po_d2v*>(p)); }
__device__ __forceinline__ void po_store2(float* p, double a, double b) { const po_f2v v = {(float)a, (float)b}; __builtin_nontemporal_store(v, reinterpret_cast<po_f2v*>(p)); }
#endif

__device__ __forceinline__ bool po_in_block(const po_tile_args& A, uint64_t i, uint64_t j) {
    return i >= A.row_begin && i < A.row_end && j >= A.col_begin && j < A.col_end;
}

// out[(i-row_begin)*ld + (j-col_begin)] and, when asked, mirror[(j-col_begin)*ldm + (i-row_begin)]
template <typename OUT>
__device__ __forceinline__ void po_store_pair(const po_tile_args& A, uint64_t i, uint64_t j, double v, bool mirror) {
    po_out_store(&static_cast<OUT*>(A.out)[(i - A.row_begin) * A.ld_out + (j - A.col_begin)], (OUT)v);
    if (mirror) po_out_store(&static_cast<OUT*>(A.mirror)[(j - A.col_begin) * A.ld_mirror + (i - A.row_begin)], (OUT)v);
}

// a tile writes its transpose iff the block has a mirror target and the tile is not on the diagonal
// of a triangular block (those tiles are computed in full)
// JSD tile kernels: `lowhi` = the smallest high word among a lane's values of this tile (>= 0, the diagonal left out); po_fix_hits
// = the lanes that hold one below 2^-20 (a wave mask, scalar registers).  If any does, one lane appends (tile, first row, rows) to the
// list of po_jsd_exact.hip.
__device__ __forceinline__ unsigned long long po_fix_hits(uint32_t lowhi) { return __builtin_amdgcn_ballot_w64(lowhi < PO_FIX_BELOW_HI); }
__device__ __forceinline__ void po_fix_note(po_fix_list* fix, unsigned long long hit, uint32_t ti, uint32_t tj, uint32_t row0, uint32_t rows) {
    if (fix == nullptr || hit == 0ull) return;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(hit)) {
        const uint32_t at = atomicAdd(&fix->count, 1u);
        if (at < PO_FIX_CAP)
            fix->entry[at] = ((unsigned long long)ti << 40) | ((unsigned long long)tj << 16) | ((unsigned long long)row0 << 8) | rows;
    }
}

__device__ __forceinline__ bool po_tile_mirrors(const po_tile_args& A, uint32_t ti, uint32_t tj) {
    return A.mirror != nullptr && !(A.triangular && ti == tj);
}
// ---- float32 output through a workgroup-wide LDS tile (round 5) ---------------------------------------
// A float32 matrix row is half as long in bytes as a float64 one, so an epilogue that stores straight from the register layout
// of its arithmetic leaves 128-byte crumbs (32 accumulator columns x 4 B) - one store instruction, one address computation and
// one bounds test per element - where the float64 path writes 256 B.  Every float32 epilogue instead puts the tile's 128 x 128
// values into LDS as float32 (tl[r * kF32TileStride + c], 65 KiB), and after ONE LDS barrier the waves write the tile and its
// transpose as 16-byte stores, two 512-byte row pieces per instruction (tools/ubench/write_bw_f32.hip: 5.07 TB/s against 4.03
// for the 128-byte crumbs when the rows of the matrix are only 64-byte aligned, as at N = 50 000).  Interior tiles of an aligned
// matrix take a path without any per-lane test; everything else (edge tiles, odd leading dimensions, blocks that start inside
// a tile) stores element by element, 256 contiguous bytes per instruction.
// Row stride 130 floats: rows stay 8-byte aligned (the direct rows are read as two ds_read_b64 per lane) and a COLUMN of the
// tile - what the transposed rows read - spreads over 8 bank groups (4-way conflicts on 4-byte reads, a few hundred LDS cycles
// per tile; a stride of 132 would make that 8-way, 136 16-way).
constexpr int kF32TileStride = 130;
constexpr int kF32TileBytes = 128 * kF32TileStride * 4;              // 66 560 B

typedef float po_f4v __attribute__((ext_vector_type(4)));

// VW consecutive entries of the scratch as one store: 16, 8 or 4 bytes
template <typename T, int VW> struct po_vec;
template <> struct po_vec<float, 4> { typedef po_f4v type; };
template <> struct po_vec<float, 2> { typedef po_f2v type; };
template <> struct po_vec<float, 1> { typedef float type; };
template <> struct po_vec<double, 2> { typedef po_d2v type; };
template <> struct po_vec<double, 1> { typedef double type; };

// A TR x TC piece of the scratch (entries of type T, row stride STRIDE entries) that lies inside its block, as stores of VW entries: a row
// of the piece in the matrix is WIDTH / VW lanes, a wave instruction covers 64 VW / WIDTH rows (float, VW = 4, WIDTH = 128: two rows,
// 16-byte stores of 512-byte pieces).  `dst` = the matrix entry of (row 0, column 0) of the piece; TRANSPOSED = false: the piece as it
// lies in the scratch (TR rows of TC entries), true: its transpose (TC rows of TR entries: the rows of the piece are columns of the scratch).
template <typename T, int NW, int TR, int TC, int STRIDE, int VW, bool TRANSPOSED>
__device__ __forceinline__ void po_store_piece(T* dst, uint64_t ld, uint32_t wave, uint32_t lane, const T* tl) {
    typedef typename po_vec<T, VW>::type vec;
    constexpr int WIDTH = TRANSPOSED ? TR : TC, NROWS = TRANSPOSED ? TC : TR;        // shape of the piece in the matrix
    constexpr int LPR = WIDTH / VW;                                  // lanes per row
    if constexpr (LPR <= 64) {
        constexpr int RPI = 64 / LPR, IT = NROWS / RPI / NW;          // rows per instruction, instructions per wave
        static_assert(IT >= 1 && IT * RPI * NW == NROWS, "the waves share the rows of a piece evenly");
        const uint32_t q = lane / LPR, m = lane % LPR;
        const uint32_t r0 = RPI * (wave & (NW - 1)) + q;
        T* out = dst + (uint64_t)r0 * ld + VW * m;
        const uint64_t step = (uint64_t)RPI * NW * ld;
        vec v[IT];
        // (all LDS reads of the wave first, then its stores; the wave index is known to be below NW, so the trip count is fixed)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const uint32_t r = r0 + it * RPI * NW;
            if constexpr (!TRANSPOSED) {
                const T* src = tl + r * STRIDE + VW * m;
                if constexpr (sizeof(T) == 4 && VW == 4) { const float2 a = *reinterpret_cast<const float2*>(src), b = *reinterpret_cast<const float2*>(src + 2); v[it] = vec{a.x, a.y, b.x, b.y}; }
                else if constexpr (sizeof(T) == 4 && VW == 2) { const float2 a = *reinterpret_cast<const float2*>(src); v[it] = vec{a.x, a.y}; }
                else if constexpr (VW == 2) v[it] = vec{src[0], src[1]};
                else v[it] = src[0];
            } else {
                const T* src = tl + VW * m * STRIDE + r;
                if constexpr (VW == 4) v[it] = vec{src[0], src[STRIDE], src[2 * STRIDE], src[3 * STRIDE]};
                else if constexpr (VW == 2) v[it] = vec{src[0], src[STRIDE]};
                else v[it] = src[0];
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) __builtin_nontemporal_store(v[it], reinterpret_cast<vec*>(out + it * step));
    } else {                                                          // 128 single entries per row: two instructions per row
        static_assert(WIDTH == 128 && VW == 1, "single entries: 64 or 128 per row");
        constexpr int IT = NROWS / NW, GRP = IT < 8 ? IT : 8;         // (eight rows of reads, then their stores)
        const uint32_t r0 = wave & (NW - 1);
        T* out = dst + (uint64_t)r0 * ld + lane;
        const uint64_t step = (uint64_t)NW * ld;
#pragma unroll 1
        for (int g = 0; g < IT; g += GRP) {
            T v[GRP][2];
#pragma unroll
            for (int it = 0; it < GRP; ++it) {
                const uint32_t r = r0 + (g + it) * NW;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    v[it][h] = TRANSPOSED ? tl[(lane + 64 * h) * STRIDE + r] : tl[r * STRIDE + lane + 64 * h];
            }
#pragma unroll
            for (int it = 0; it < GRP; ++it) {
                __builtin_nontemporal_store(v[it][0], out + (g + it) * step);
                __builtin_nontemporal_store(v[it][1], out + (g + it) * step + 64);
            }
        }
    }
}

// The widest store a piece allows, in entries: 16 bytes when its rows start on 16-byte boundaries, 8 bytes on 8-byte boundaries, else
// (float only) single entries
template <typename T>
__device__ __forceinline__ int po_store_width(const void* base, uint64_t ld, uint64_t first) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    constexpr uint32_t per16 = 16 / sizeof(T), per8 = 8 / sizeof(T);
    if ((ld % per16) == 0 && (first % per16) == 0 && (a & 15) == 0) return (int)per16;
    if ((ld % per8) == 0 && (first % per8) == 0 && (a & 7) == 0) return (int)per8;
    return 1;
}

// All NW waves of the workgroup call this after the values sit in `tl` and a barrier has made them visible.  The scratch holds TR rows
// (i0 ..) of TC columns (j0 ..) of the tile, row stride STRIDE entries of type T (the output type): the whole 128 x 128 tile, a band of
// its rows (TR = 64 or 32: a kernel that keeps a part of the scratch and makes several passes; its transposed rows then leave as shorter
// pieces) or its left or right half (TC = 64).  Tiles inside their block go out without a per-lane test, as 16-byte stores when the rows
// of the matrix start on 16-byte boundaries (float32: a leading dimension that is a multiple of 4 entries; of 32 entries = whole
// 128-byte lines is better still, and is what this library's own buffers have), as 8- or 4-byte stores otherwise.
template <typename T, int NW, int TR, int TC, int STRIDE>
__device__ __forceinline__ void po_store_tile(const po_tile_args& A, bool mirrors, uint64_t i0, uint64_t j0, uint32_t wave,
                                              uint32_t lane, const T* tl) {
    static_assert((TR == 128 || TR == 64 || TR == 32) && (TC == 128 || TC == 64) && STRIDE >= TC && STRIDE % 2 == 0, "whole tiles or parts");
    T* out = static_cast<T*>(A.out);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    const bool inside = i0 >= A.row_begin && i0 + TR <= n_rows && j0 >= A.col_begin && j0 + TC <= n_cols;   // uniform
    constexpr int VMAX = 16 / sizeof(T);
    // ---- the rows themselves ----
    if (inside) {
        T* dst = out + (i0 - A.row_begin) * A.ld_out + (j0 - A.col_begin);
        const int vw = po_store_width<T>(A.out, A.ld_out, j0 - A.col_begin);
        if (vw == VMAX) po_store_piece<T, NW, TR, TC, STRIDE, VMAX, false>(dst, A.ld_out, wave, lane, tl);
        else if (vw == VMAX / 2) po_store_piece<T, NW, TR, TC, STRIDE, VMAX / 2, false>(dst, A.ld_out, wave, lane, tl);
        else if constexpr (VMAX == 4) po_store_piece<T, NW, TR, TC, STRIDE, 1, false>(dst, A.ld_out, wave, lane, tl);
    } else {
        for (uint32_t r = wave; r < TR; r += NW) {
            const uint64_t i = i0 + r;
            if (i < A.row_begin || i >= n_rows) continue;
            T* row = out + (i - A.row_begin) * A.ld_out;
#pragma unroll
            for (uint32_t c = lane; c < TC; c += 64) {
                const uint64_t j = j0 + c;
                if (j >= A.col_begin && j < n_cols) po_out_store(&row[j - A.col_begin], tl[r * STRIDE + c]);
            }
        }
    }
    if (!mirrors) return;
    // ---- the transposed rows (columns of tl) ----
    T* mir = static_cast<T*>(A.mirror);
    if (inside) {
        T* dst = mir + (j0 - A.col_begin) * A.ld_mirror + (i0 - A.row_begin);
        const int vw = po_store_width<T>(A.mirror, A.ld_mirror, i0 - A.row_begin);
        if (vw == VMAX) po_store_piece<T, NW, TR, TC, STRIDE, VMAX, true>(dst, A.ld_mirror, wave, lane, tl);
        else if (vw == VMAX / 2) po_store_piece<T, NW, TR, TC, STRIDE, VMAX / 2, true>(dst, A.ld_mirror, wave, lane, tl);
        else if constexpr (VMAX == 4) po_store_piece<T, NW, TR, TC, STRIDE, 1, true>(dst, A.ld_mirror, wave, lane, tl);
    } else {
        for (uint32_t c = wave; c < TC; c += NW) {
            const uint64_t j = j0 + c;
            if (j < A.col_begin || j >= n_cols) continue;
            T* row = mir + (j - A.col_begin) * A.ld_mirror;
#pragma unroll
            for (uint32_t r = lane; r < TR; r += 64) {
                const uint64_t i = i0 + r;
                if (i >= A.row_begin && i < n_rows) po_out_store(&row[i - A.row_begin], tl[r * STRIDE + c]);
            }
        }
    }
}

// the float32 form by its round-5 name
template <int NW, int TR = 128, int TC = 128, int STRIDE = kF32TileStride>
__device__ __forceinline__ void po_store_tile_f32(const po_tile_args& A, bool mirrors, uint64_t i0, uint64_t j0, uint32_t wave,
                                                  uint32_t lane, const float* tl) {
    po_store_tile<float, NW, TR, TC, STRIDE>(A, mirrors, i0, j0, wave, lane, tl);
}

// ---- register-block epilogue of the VALU tile kernels -------------------------------------------------
// Lane (tx, ty) of an NT-lane workgroup holds v[ia][ib] for record rows i0 + ty*RPT + ia and record
// columns j0 + 32*(ib>>1) + 2*tx + (ib&1).  The tile goes out as 16-byte stores along rows (16 lanes =
// 256 contiguous bytes); the mirrored tile is transposed through LDS, 32 columns at a time, so that it
// too leaves as full contiguous row segments (one wave = one 1 KiB row piece) instead of 64-byte crumbs.

constexpr int kMirrorLdsStride = 130;                                 // doubles per transposed row (128 + pad)
constexpr int kMirrorLdsBytes = 32 * kMirrorLdsStride * 8;           // 33 280 B of LDS scratch

// One column group q (record columns j0 + 32 q + 2 tx + {0, 1}) of the register block: v[ia][e].  The direct tile piece is
// stored, then - when the tile has a mirror - the 32 transposed rows of the group go through LDS and out.  Calling this for
// q = 0..3 is the whole epilogue; a kernel that produces its values group by group (valu_tile_kernel) calls it as it goes, so
// that a group's registers are dead before the next group's per-record terms arrive.
template <typename OUT, int RPT, int NT>
__device__ __forceinline__ void po_store_block_part(const po_tile_args& A, bool mirrors, uint64_t i0, uint64_t j0, uint32_t tx, uint32_t ty,
                                                    int q, const double (&v)[RPT][2], double* lds) {
    OUT* out = static_cast<OUT*>(A.out);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    // ---- the tile itself: two adjacent columns per lane leave as one 16-byte (float64) or 8-byte (float32) store ----
    const bool vec_out = (A.ld_out & 1) == 0 && ((j0 - A.col_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.out) & (2 * sizeof(OUT) - 1)) == 0 && j0 >= A.col_begin;
    const uint64_t j = j0 + 32 * q + 2 * tx;
#pragma unroll
    for (int ia = 0; ia < RPT; ++ia) {
        const uint64_t i = i0 + ty * RPT + ia;
        if (i < A.row_begin || i >= n_rows) continue;
        OUT* row = out + (i - A.row_begin) * A.ld_out;
        if (vec_out && j + 1 < n_cols) {
            po_store2(row + (j - A.col_begin), v[ia][0], v[ia][1]);
        } else {
            if (j >= A.col_begin && j < n_cols) po_out_store(&row[j - A.col_begin], (OUT)v[ia][0]);
            if (j + 1 >= A.col_begin && j + 1 < n_cols) po_out_store(&row[j + 1 - A.col_begin], (OUT)v[ia][1]);
        }
    }
    if (!mirrors) return;                                             // uniform over the workgroup
    // ---- the transposed rows of this group ----
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint32_t t = ty * 16 + tx, lane = t & 63, wave = t >> 6;
    const bool vec_mir = (A.ld_mirror & 1) == 0 && ((i0 - A.row_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.mirror) & (2 * sizeof(OUT) - 1)) == 0 && i0 >= A.row_begin;
    po_lds_barrier();                                                 // LDS only: the stores above stay in flight
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ia = 0; ia < RPT; ++ia) lds[(2 * tx + e) * kMirrorLdsStride + ty * RPT + ia] = v[ia][e];
    po_lds_barrier();
    for (uint32_t jq = wave; jq < 32; jq += NT / 64) {               // one wave per transposed row
        const uint64_t jr = j0 + 32 * q + jq;
        if (jr < A.col_begin || jr >= n_cols) continue;
        const double2 w = *reinterpret_cast<const double2*>(lds + jq * kMirrorLdsStride + 2 * lane);
        const uint64_t i = i0 + 2 * lane;
        OUT* row = mir + (jr - A.col_begin) * A.ld_mirror;
        if (vec_mir && i + 1 < n_rows) {
            po_store2(row + (i - A.row_begin), w.x, w.y);
        } else {
            if (i >= A.row_begin && i < n_rows) po_out_store(&row[i - A.row_begin], (OUT)w.x);
            if (i + 1 >= A.row_begin && i + 1 < n_rows) po_out_store(&row[i + 1 - A.row_begin], (OUT)w.y);
        }
    }
}

// float32 output of the same register block (round 5): instead of 8-byte stores of two columns per lane - 128-byte row pieces -
// the values go through the float32 scratch of po_store_tile_f32, half a tile at a time (rows 0..63 are held by the lanes with
// ty * RPT < 64, rows 64..127 by the others; 33 KiB, the size of the mirror scratch above), and leave as 16-byte stores of
// 512-byte (tile) and 256-byte (transposed tile) row pieces.
template <int RPT, int NT>
__device__ __forceinline__ void po_store_block_f32(const po_tile_args& A, uint32_t ti, uint32_t tj, uint64_t i0, uint64_t j0,
                                                   uint32_t tx, uint32_t ty, const double (&v)[RPT][8], float* tl) {
    const bool mirrors = po_tile_mirrors(A, ti, tj);                 // uniform over the workgroup
    const uint32_t t = ty * 16 + tx, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        po_lds_barrier();                                             // whoever read the scratch before is done
        if ((ty * RPT) / 64 == h) {
            float* row = tl + (ty * RPT - 64 * h) * kF32TileStride + 2 * tx;
#pragma unroll
            for (int ia = 0; ia < RPT; ++ia)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float2*>(row + ia * kF32TileStride + 32 * q) = make_float2((float)v[ia][2 * q], (float)v[ia][2 * q + 1]);
        }
        po_lds_barrier();
        po_store_tile_f32<NT / 64, 64, 128>(A, mirrors, i0 + 64 * h, j0, wave, lane, tl);
    }
}

template <typename OUT, int RPT, int NT>
__device__ __forceinline__ void po_store_block(const po_tile_args& A, uint32_t ti, uint32_t tj, uint64_t i0, uint64_t j0,
                                               uint32_t tx, uint32_t ty, const double (&v)[RPT][8], double* lds) {
    OUT* out = static_cast<OUT*>(A.out);
    const uint64_t n_rows = min(A.row_end, A.n), n_cols = min(A.col_end, A.n);
    // ---- the tile itself ----
    // two adjacent columns per lane leave as one 16-byte (float64) or 8-byte (float32) store
    const bool vec_out = (A.ld_out & 1) == 0 && ((j0 - A.col_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.out) & (2 * sizeof(OUT) - 1)) == 0 && j0 >= A.col_begin;
#pragma unroll
    for (int ia = 0; ia < RPT; ++ia) {
        const uint64_t i = i0 + ty * RPT + ia;
        if (i < A.row_begin || i >= n_rows) continue;
        OUT* row = out + (i - A.row_begin) * A.ld_out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t j = j0 + 32 * q + 2 * tx;
            if (vec_out && j + 1 < n_cols) {
                po_store2(row + (j - A.col_begin), v[ia][2 * q], v[ia][2 * q + 1]);
            } else {
                if (j >= A.col_begin && j < n_cols) po_out_store(&row[j - A.col_begin], (OUT)v[ia][2 * q]);
                if (j + 1 >= A.col_begin && j + 1 < n_cols) po_out_store(&row[j + 1 - A.col_begin], (OUT)v[ia][2 * q + 1]);
            }
        }
    }
    if (!po_tile_mirrors(A, ti, tj)) return;                          // uniform over the workgroup
    // ---- the transposed tile ----
    OUT* mir = static_cast<OUT*>(A.mirror);
    const uint32_t t = ty * 16 + tx, lane = t & 63, wave = t >> 6;
    const bool vec_mir = (A.ld_mirror & 1) == 0 && ((i0 - A.row_begin) & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(A.mirror) & (2 * sizeof(OUT) - 1)) == 0 && i0 >= A.row_begin;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        po_lds_barrier();                                             // LDS only: the stores above stay in flight
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ia = 0; ia < RPT; ++ia) lds[(2 * tx + e) * kMirrorLdsStride + ty * RPT + ia] = v[ia][2 * q + e];
        po_lds_barrier();
        for (uint32_t jq = wave; jq < 32; jq += NT / 64) {           // one wave per transposed row
            const uint64_t j = j0 + 32 * q + jq;
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

#endif
