// Micro-benchmark: cycles per v_mfma_scale_f32_32x32x64_f8f6f4 (both operands FP4) and per v_mfma_i32_32x32x32_i8 on gfx950, in the
// register pattern of pairdot_tile_kernel (8 accumulators of 32 x 32, 4 + 2 operand fragments), with s_memtime around the loop:
//   MODE 0  matrix instructions only (operands stay in registers)            1 or 2 waves per SIMD
//   MODE 1  + the 6 ds_read_b128 of a k-step ahead of its 8 matrix instructions (operands re-read from LDS every k-step)
// Prints cycles per matrix instruction per SIMD and the clock the chip held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int FP4, int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int iters, const uint4* seed) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (uint32_t i = t; i < 32768 / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = seed[i & 1023];
    __syncthreads();
    v16f gf[4][2];
    v16i gi[4][2];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) { gf[m][n][e] = 0; gi[m][n][e] = 0; }
    v4i a[4], b[2];
    for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4i*>(smem + ((wv & 1) * 128 + m * 32 + (lane & 31)) * 16);
    for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const v4i*>(smem + (256 + (wv >> 1) * 64 + n * 32 + (lane & 31)) * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const unsigned char* base = smem + ((it & 3) * 2 + (lane >> 5)) * 4096;
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4i*>(base + ((wv & 1) * 64 + m * 16 + (lane & 15)) * 16);
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = *reinterpret_cast<const v4i*>(base + (128 + (wv >> 1) * 32 + n * 16 + (lane & 15)) * 16);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (FP4) {
                    const v8i a8 = {a[m][0], a[m][1], a[m][2], a[m][3], 0, 0, 0, 0};
                    const v8i b8 = {b[n][0], b[n][1], b[n][2], b[n][3], 0, 0, 0, 0};
                    gf[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, gf[m][n], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                } else {
                    gi[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], gi[m][n], 0, 0, 0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) s += gf[m][n][e] + (float)gi[m][n][e];
    if (lane == 0) { out[(blockIdx.x * 8 + wv) * 3] = t1 - t0; out[(blockIdx.x * 8 + wv) * 3 + 1] = r1 - r0; out[(blockIdx.x * 8 + wv) * 3 + 2] = (unsigned long long)s; }
}

template <int FP4, int MODE> void run(const char* name, int waves_per_simd, unsigned long long* d, const uint4* seed) {
    const int iters = 20000, blocks = 256 * 4;
    const int threads = 256 * waves_per_simd;                 // one workgroup per CU (LDS 96 KiB keeps a second one out)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<FP4, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL((k<FP4, MODE>), dim3(blocks), dim3(threads), 98304, 0, d, 200, seed);
    hipLaunchKernelGGL((k<FP4, MODE>), dim3(blocks), dim3(threads), 98304, 0, d, iters, seed);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 8 * 3);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0; int cnt = 0;
    for (int bq = 0; bq < blocks; ++bq) for (int w = 0; w < threads / 64; ++w) {
        const unsigned long long c = h[(bq * 8 + w) * 3], r = h[(bq * 8 + w) * 3 + 1];
        cyc += (double)c; ghz += (double)c / ((double)r * 10.0); ++cnt;
    }
    cyc /= cnt; ghz /= cnt;
    printf("%-58s %d wave(s)/SIMD: %.1f cycles per matrix instruction per SIMD, clock %.2f GHz\n", name, waves_per_simd,
           cyc / ((double)iters * 8 * waves_per_simd), ghz);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 1 << 20);
    std::vector<uint32_t> hs(4096);
    for (size_t i = 0; i < hs.size(); ++i) { uint32_t x = (uint32_t)i * 2654435761u; uint32_t w = 0; for (int e = 0; e < 8; ++e) { const uint32_t r = (x >> (3 * e)) % 3; w |= (r == 0 ? 0x0u : r == 1 ? 0x2u : 0xAu) << (4 * e); } hs[i] = w; }
    uint4* seed; hipMalloc(&seed, 16384); hipMemcpy(seed, hs.data(), 16384, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>("FP4 32x32x64, operands in registers", w, d, seed);
        run<1, 1>("FP4 32x32x64, 6 ds_read_b128 per 8 matrix instructions", w, d, seed);
        run<0, 0>("int8 32x32x32, operands in registers", w, d, seed);
        run<0, 1>("int8 32x32x32, 6 ds_read_b128 per 8 matrix instructions", w, d, seed);
    }
    return 0;
}
