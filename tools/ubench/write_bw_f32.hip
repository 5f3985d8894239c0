// Achievable HBM store bandwidth for a FLOAT32 distance matrix (the container type of --large memmap / h5py and of every
// multi-GPU CLI run): upper-triangle 128 x 128 tiles in the XCD-banded order of po_tiles.h, each tile stored twice (itself and
// its transpose), non-temporal stores, for the shapes a store instruction of a tile epilogue can have:
//   mode 0   4-byte stores, 2 rows x 128 B per instruction   (the 32 x 32 accumulator layout as it falls out: round 4's float32 path)
//   mode 1   4-byte stores, 1 row  x 256 B
//   mode 2   8-byte stores, 2 rows x 256 B
//   mode 3   8-byte stores, 1 row  x 512 B
//   mode 4  16-byte stores, 2 rows x 512 B
//   mode 5  16-byte stores, 4 rows x 256 B
//   mode 6  16-byte stores, 1 row  x 1 KiB  (256 x 256 tiles)
//   mode 7   8-byte stores, 4 rows x 128 B
// argv[2] = matrix size (default 49 920);  argv[1] = leading dimension in elements (default 50 048: rows 128-byte aligned; 50 000 puts every other row 64 bytes off)
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude -Iphyloligo_amd/csrc -o tools/ubench/write_bw_f32 tools/ubench/write_bw_f32.hip
#include "po_tiles.h"
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill_tri_f32(po_tile_args A, float* __restrict__ out, uint64_t ld) {
    constexpr uint32_t E = MODE == 6 ? 256 : 128;
    uint32_t ti, tj;
    po_tile_coords(A, E, blockIdx.x, ti, tj);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 0; o < 2; ++o) {
        if (o == 1 && ti == tj) break;
        const uint64_t i0 = (uint64_t)(o ? tj : ti) * E, j0 = (uint64_t)(o ? ti : tj) * E;
        float* base = out + i0 * ld + j0;
        if (MODE == 0) {                                   // wave w: rows 32 w .. 32 w + 31, four 32-column blocks
            const uint32_t lr = lane & 31, lh = lane >> 5;
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    __builtin_nontemporal_store((float)reg, base + (uint64_t)(w * 32 + rl) * ld + cb * 32 + lr);
                }
        } else if (MODE == 1) {
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll 8
                for (int r = 0; r < 32; ++r) __builtin_nontemporal_store((float)r, base + (uint64_t)(w * 32 + r) * ld + cb * 64 + lane);
        } else if (MODE == 2) {                            // lanes 0..31 one row (64 columns as float2), lanes 32..63 the row 4 below
            const uint32_t lr = lane & 31, lh = lane >> 5;
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t rl = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    const f2v v = {(float)reg, (float)lane};
                    __builtin_nontemporal_store(v, reinterpret_cast<f2v*>(base + (uint64_t)(w * 32 + rl) * ld + cb * 64 + 2 * lr));
                }
        } else if (MODE == 3) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const f2v v = {(float)r, (float)lane};
                __builtin_nontemporal_store(v, reinterpret_cast<f2v*>(base + (uint64_t)(w * 32 + r) * ld + 2 * lane));
            }
        } else if (MODE == 4) {
#pragma unroll 8
            for (int r = 0; r < 32; r += 2) {
                const f4v v = {(float)r, (float)lane, 1.f, 2.f};
                __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(base + (uint64_t)(w * 32 + r + (lane >> 5)) * ld + 4 * (lane & 31)));
            }
        } else if (MODE == 5) {
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll 8
                for (int r = 0; r < 32; r += 4) {
                    const f4v v = {(float)r, (float)lane, 1.f, 2.f};
                    __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(base + (uint64_t)(w * 32 + r + (lane >> 4)) * ld + cb * 64 + 4 * (lane & 15)));
                }
        } else if (MODE == 6) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                const f4v v = {(float)r, (float)lane, 1.f, 2.f};
                __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(base + (uint64_t)(w * 64 + r) * ld + 4 * lane));
            }
        } else {
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll 8
                for (int r = 0; r < 32; r += 4) {
                    const f2v v = {(float)r, (float)lane};
                    __builtin_nontemporal_store(v, reinterpret_cast<f2v*>(base + (uint64_t)(w * 32 + r + (lane >> 4)) * ld + cb * 32 + 2 * (lane & 15)));
                }
        }
    }
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 10) : 49920;   // a multiple of 256 (49 920 = 390 tiles of 128; 199 936 = config 4's size, 160 GB)
    const uint64_t ld = argc > 1 ? strtoull(argv[1], nullptr, 10) : 50048;
    float* out;
    if (hipMalloc(&out, (size_t)n * ld * 4 + 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    po_tile_args A;
    memset(&A, 0, sizeof(A));
    A.n = A.npad = n; A.row_end = A.col_end = n; A.triangular = 1;
    static const char* names[8] = {"4-byte stores, 2 rows x 128 B", "4-byte stores, 1 row x 256 B", "8-byte stores, 2 rows x 256 B",
                                   "8-byte stores, 1 row x 512 B", "16-byte stores, 2 rows x 512 B", "16-byte stores, 4 rows x 256 B",
                                   "16-byte stores, 1 row x 1 KiB (256 x 256 tiles)", "8-byte stores, 4 rows x 128 B"};
    printf("float32 matrix %u x %u, leading dimension %llu (%s rows), triangle + mirror, XCD-banded order, nt stores\n", n, n,
           (unsigned long long)ld, (ld * 4) % 128 ? "64-byte aligned" : "128-byte aligned");
    for (int mode = 0; mode < 8; ++mode) {
        const uint32_t T = n / (mode == 6 ? 256 : 128);
        const dim3 grid(T * (T + 1) / 2);
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(fill_tri_f32<0>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 1: hipLaunchKernelGGL(fill_tri_f32<1>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 2: hipLaunchKernelGGL(fill_tri_f32<2>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 3: hipLaunchKernelGGL(fill_tri_f32<3>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 4: hipLaunchKernelGGL(fill_tri_f32<4>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 5: hipLaunchKernelGGL(fill_tri_f32<5>, grid, dim3(256), 0, 0, A, out, ld); break;
                case 6: hipLaunchKernelGGL(fill_tri_f32<6>, grid, dim3(256), 0, 0, A, out, ld); break;
                default: hipLaunchKernelGGL(fill_tri_f32<7>, grid, dim3(256), 0, 0, A, out, ld); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("  mode %d  %-48s %.3f ms  %.2f TB/s\n", mode, names[mode], best, (double)n * n * 4 / (best * 1e-3) / 1e12);
    }
    hipFree(out);
    return 0;
}
