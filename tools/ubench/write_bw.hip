// Achievable HBM store bandwidth on this GPU for the output pattern of the stage-2 kernels:
// an N x N float64 matrix written once, (a) as a flat fill with 16-byte stores, (b) tile by tile
// (128 x 128 tiles, 16-byte stores along rows = 1 KiB row pieces, XCD-interleaved tile order) as po_store_block does.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/write_bw tools/ubench/write_bw.hip
// (c) the stage-2 pattern proper: upper-triangle tiles in the XCD-aware banded order of po_tiles.h, each tile
// stored twice (itself and its transpose), still as 1 KiB row pieces.
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude -Iphyloligo_amd/csrc -o tools/ubench/write_bw tools/ubench/write_bw.hip
#include "po_tiles.h"
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void fill_tri(po_tile_args A, double* __restrict__ out, uint32_t n) {
    uint32_t ti, tj;
    po_tile_coords(A, 128, blockIdx.x, ti, tj);
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int o = 0; o < 2; ++o) {
        if (o == 1 && ti == tj) break;
        const uint32_t a = o ? tj : ti, b = o ? ti : tj;
        for (uint32_t r = ty; r < 128; r += 4) {
            const uint64_t i = (uint64_t)a * 128 + r, j = (uint64_t)b * 128 + 2 * tx;
            *reinterpret_cast<double2*>(out + i * n + j) = make_double2((double)i, (double)j);
        }
    }
}

__global__ __launch_bounds__(256) void fill_flat(double2* __restrict__ out, uint64_t n2) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * 256)
        out[i] = make_double2((double)i, 1.0);
}

__global__ __launch_bounds__(256) void fill_tiles(double* __restrict__ out, uint32_t n, uint32_t tiles_per_row) {
    const uint32_t ti = blockIdx.x / tiles_per_row, tj = blockIdx.x % tiles_per_row;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;           // one wave = one 1 KiB row piece
    for (uint32_t r = ty; r < 128; r += 4) {
        const uint64_t i = (uint64_t)ti * 128 + r, j = (uint64_t)tj * 128 + 2 * tx;
        if (i < n && j + 1 < n) *reinterpret_cast<double2*>(out + i * n + j) = make_double2((double)i, (double)j);
    }
}

// (d) the matrix-core kernels' pattern: 8-byte stores, one instruction = 2 rows x 32 columns (two 256-byte segments), as the
// 32x32 accumulator layout gives them (column = lane & 31, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
// (e) the same bytes as one row x 64 columns per instruction (512 contiguous bytes).
template <int WIDE>
__global__ __launch_bounds__(256) void fill_tri_acc(po_tile_args A, double* __restrict__ out, uint32_t n) {
    uint32_t ti, tj;
    po_tile_coords(A, 128, blockIdx.x, ti, tj);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    for (int o = 0; o < 2; ++o) {
        if (o == 1 && ti == tj) break;
        const uint64_t i0 = (uint64_t)(o ? tj : ti) * 128 + w * 32, j0 = (uint64_t)(o ? ti : tj) * 128;
        if (!WIDE) {
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    out[(i0 + rl) * n + j0 + cb * 32 + lr] = (double)reg;
                }
        } else {
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 32; ++r) out[(i0 + r) * n + j0 + cb * 64 + lane] = (double)r;
        }
    }
}

// (f) mirror written as 128-byte pieces: one 16-byte store instruction = 8 rows x 128 B (8 lanes per row piece), non-temporal
__global__ __launch_bounds__(256) void fill_tri_128(po_tile_args A, double* __restrict__ out, uint32_t n) {
    uint32_t ti, tj;
    po_tile_coords(A, 128, blockIdx.x, ti, tj);
    typedef double d2v __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 0; o < 2; ++o) {
        if (o == 1 && ti == tj) break;
        const uint64_t i0 = (uint64_t)(o ? tj : ti) * 128, j0 = (uint64_t)(o ? ti : tj) * 128;
        // wave w owns the 16-column strip [32 w' ...): rows r, columns j0 + 16 * (w * 2 + h) + 2 * (lane & 7)
        for (int h = 0; h < 2; ++h)
            for (uint32_t r = lane >> 3; r < 128; r += 8) {
                const d2v v = {(double)r, (double)lane};
                __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(out + (i0 + r) * n + j0 + 16 * (w * 2 + h) + 2 * (lane & 7)));
            }
    }
}

int main() {
    const uint32_t n = 50048;                      // multiple of 128
    double* out;
    hipMalloc(&out, (size_t)n * n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    po_tile_args A;
    memset(&A, 0, sizeof(A));
    A.n = A.npad = n; A.row_end = A.col_end = n; A.triangular = 1;
    const uint32_t T = n / 128;
    for (int mode = 0; mode < 6; ++mode) {
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(fill_flat, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<double2*>(out), (uint64_t)n * n / 2);
            else if (mode == 5) hipLaunchKernelGGL(fill_tri_128, dim3(T * (T + 1) / 2), dim3(256), 0, 0, A, out, n);
            else if (mode == 3) hipLaunchKernelGGL(fill_tri_acc<0>, dim3(T * (T + 1) / 2), dim3(256), 0, 0, A, out, n);
            else if (mode == 4) hipLaunchKernelGGL(fill_tri_acc<1>, dim3(T * (T + 1) / 2), dim3(256), 0, 0, A, out, n);
            else if (mode == 2) hipLaunchKernelGGL(fill_tri, dim3(T * (T + 1) / 2), dim3(256), 0, 0, A, out, n);
            else hipLaunchKernelGGL(fill_tiles, dim3((n / 128) * (n / 128)), dim3(256), 0, 0, out, n, n / 128);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms  %.2f TB/s\n", mode == 0 ? "flat 16-byte fill" : mode == 1 ? "128x128 tiles, 1 KiB row pieces" : mode == 2 ? "triangle + mirror, XCD-banded order" : mode == 3 ? "triangle + mirror, 8-byte stores, 2 rows x 256 B per instruction" : mode == 4 ? "triangle + mirror, 8-byte stores, 1 row x 512 B per instruction" : "triangle + mirror, 16-byte nt stores, 8 rows x 128 B per instruction", ms,
                   (double)n * n * 8 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
