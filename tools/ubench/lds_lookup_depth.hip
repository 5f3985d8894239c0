// Micro-benchmark: ds_read_b64 table lookups (every lane its own bank-distinct copy of a 64 KiB table, random entries - the access
// pattern of the JSD table kernel) against the number of lookups a wave keeps in flight and the waves per SIMD.  Cycles of the CU's
// LDS data path per wave-instruction = time * clock / (lookups issued per CU); clock from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int DEPTH, int SEQ>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (uint32_t i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (SEQ ? lane * 8 : (lane & 31) * 8);
    uint32_t e[16];
    for (int i = 0; i < 16; ++i) e[i] = SEQ ? ((threadIdx.x >> 6) * 16 + i) % 128 : (threadIdx.x * 7 + i * 13) % 256;
    uint32_t acc0 = 0, acc1 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const uint32_t addr = base + e[r & 15] * (SEQ ? 512 : 256);
            uint2 v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
            if (DEPTH == 2) asm volatile("s_waitcnt lgkmcnt(2)");
            if (DEPTH == 4) asm volatile("s_waitcnt lgkmcnt(4)");
            if (DEPTH == 8) asm volatile("s_waitcnt lgkmcnt(8)");
            if (DEPTH == 15) asm volatile("s_waitcnt lgkmcnt(15)");
            acc0 ^= v.x; acc1 ^= v.y;
        }
        for (int i = 0; i < 16; ++i) e[i] = (e[i] + 1 + (acc0 & 1)) % (SEQ ? 128 : 256);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 3] = t1 - t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 3 + 1] = r1 - r0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 3 + 2] = acc0 ^ acc1; }
}
template <int DEPTH, int SEQ> void run(int wg_threads, int wgs_per_cu, unsigned long long* d) {
    const int iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<DEPTH, SEQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int blocks = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<DEPTH, SEQ>), dim3(blocks), dim3(wg_threads), 65536, 0, d, 50);
    hipLaunchKernelGGL((k<DEPTH, SEQ>), dim3(blocks), dim3(wg_threads), 65536, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 16 * 3);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0; int cnt = 0;
    const int waves = wg_threads / 64;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { const double c = (double)h[(b * 16 + w) * 3], r = (double)h[(b * 16 + w) * 3 + 1]; cyc += c; ghz += c / (r * 10.0); ++cnt; }
    cyc /= cnt; ghz /= cnt;
    const double per_cu = (double)iters * 32 * waves * wgs_per_cu;            // lookups issued per CU while a wave runs its loop
    printf("%s in flight per wave %2d, %2d waves per CU: %.2f LDS cycles per wave-lookup, clock %.2f GHz\n", SEQ ? "sequential rows" : "random rows    ", DEPTH, waves * wgs_per_cu, cyc / per_cu, ghz);
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 8 << 20);
    run<2, 0>(512, 2, d); run<4, 0>(512, 2, d); run<8, 0>(512, 2, d); run<15, 0>(512, 2, d);
    run<4, 0>(1024, 2, d); run<8, 0>(1024, 2, d); run<15, 0>(1024, 2, d);
    run<8, 1>(512, 2, d); run<15, 1>(1024, 2, d);
    return 0;
}
