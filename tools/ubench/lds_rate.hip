// Micro-benchmark: LDS read throughput on gfx950 for the table-lookup access pattern of the JSD table kernel.
// 16 waves per CU (4 per SIMD), every lane reads its own bank-distinct slot of a replicated table; cycles of the CU's
// LDS data path per wave-instruction = time * clock / (instructions issued per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 32
// MODE 0: ds_read_b32, 64 copies of 4 B (entry stride 256 B)      MODE 1: ds_read_b64, 32 copies of 8 B (stride 256 B)
// MODE 2: ds_read_b128, 16 copies of 16 B (stride 256 B)          MODE 3: ds_read_b32, 32 copies of 4 B (stride 128 B)
// MODE 4: ds_read_u16 , 64 copies, 4-byte slots                   MODE 5: ds_read_b64, all lanes one address (broadcast)
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (uint32_t i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t copy;
    if (MODE == 0 || MODE == 4) copy = lane * 4;
    else if (MODE == 1) copy = (lane & 31) * 8;
    else if (MODE == 2) copy = (lane & 15) * 16;
    else if (MODE == 3) copy = (lane & 31) * 4;
    else copy = 0;
    const uint32_t stride = (MODE == 3) ? 128 : 256;
    const uint32_t nent = 65536 / stride;
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t e[8];
    for (int i = 0; i < 8; ++i) e[i] = (threadIdx.x * 7 + i * 13) % nent;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const uint32_t addr = base + copy + e[r & 7] * stride;
            if (MODE == 0 || MODE == 3) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); acc0 ^= v; }
            if (MODE == 4) { uint32_t v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); acc0 ^= v; }
            if (MODE == 1 || MODE == 5) { uint2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); acc0 ^= v.x; acc1 ^= v.y; }
            if (MODE == 2) { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)"); acc0 ^= v.x; acc1 ^= v.y; acc2 ^= v.z; acc3 ^= v.w; }
        }
        for (int i = 0; i < 8; ++i) e[i] = (e[i] + 1 + (acc0 & 1)) % nent;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}
template <int MODE> void run(const char* name, uint32_t* d) {
    const int iters = 2048; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    dim3 grid(256 * 4), block(1024);                      // 16 waves per CU at a time (64 KiB LDS each, 2 fit; registers allow it)
    hipLaunchKernelGGL(k<MODE>, grid, block, 65536, 0, d, 4); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, grid, block, 65536, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * REP * 16 * 4;   // 4 workgroups of 16 waves per CU over the launch
    printf("%-44s %.2f LDS cycles per wave-instruction (at 2.4 GHz), %.0f B/clk/CU\n", name, ms * 1e-3 * 2.4e9 / instr_per_cu,
           64.0 * (MODE == 0 || MODE == 3 ? 4 : MODE == 4 ? 2 : MODE == 2 ? 16 : 8) / (ms * 1e-3 * 2.4e9 / instr_per_cu));
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 << 20);
    run<0>("ds_read_b32  64 copies x 4 B", d);
    run<3>("ds_read_b32  32 copies x 4 B (half waves share)", d);
    run<4>("ds_read_u16  64 copies x 4 B slots", d);
    run<1>("ds_read_b64  32 copies x 8 B", d);
    run<2>("ds_read_b128 16 copies x 16 B", d);
    run<5>("ds_read_b64  one address per wave", d);
    return 0;
}
