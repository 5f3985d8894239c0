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

// absolute tile indices of workgroup `b`
__device__ __forceinline__ void po_tile_coords(const po_tile_args& A, uint32_t edge, uint64_t b, uint32_t& ti, uint32_t& tj) {
    const uint32_t r0 = (uint32_t)(A.row_begin / edge), r1 = (uint32_t)((A.row_end + edge - 1) / edge);
    const uint32_t c0 = (uint32_t)(A.col_begin / edge), c1 = (uint32_t)((A.col_end + edge - 1) / edge);
    if (A.triangular) {
        po_tri_decode(b, r1 - r0, ti, tj);
        ti += r0;
        tj += r0;
    } else {
        const uint32_t tc = c1 - c0;
        ti = r0 + (uint32_t)(b / tc);
        tj = c0 + (uint32_t)(b % tc);
    }
}

__device__ __forceinline__ bool po_in_block(const po_tile_args& A, uint64_t i, uint64_t j) {
    return i >= A.row_begin && i < A.row_end && j >= A.col_begin && j < A.col_end;
}

// out[(i-row_begin)*ld + (j-col_begin)] and, when asked, mirror[(j-col_begin)*ldm + (i-row_begin)]
template <typename OUT>
__device__ __forceinline__ void po_store_pair(const po_tile_args& A, uint64_t i, uint64_t j, double v, bool mirror) {
    static_cast<OUT*>(A.out)[(i - A.row_begin) * A.ld_out + (j - A.col_begin)] = (OUT)v;
    if (mirror) static_cast<OUT*>(A.mirror)[(j - A.col_begin) * A.ld_mirror + (i - A.row_begin)] = (OUT)v;
}

// a tile writes its transpose iff the block has a mirror target and the tile is not on the diagonal
// of a triangular block (those tiles are computed in full)
__device__ __forceinline__ bool po_tile_mirrors(const po_tile_args& A, uint32_t ti, uint32_t tj) {
    return A.mirror != nullptr && !(A.triangular && ti == tj);
}
#endif
