// Micro-benchmark: HBM write rate against the number of CUs that store at the same time (one 512-lane workgroup per CU,
// forced by a 100 KiB LDS allocation; every workgroup streams 1 MiB "tiles" of 16-byte stores, row pieces of 2 KiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512) void k(double* out, uint64_t tile_doubles, int tiles_per_wg) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    for (int t = 0; t < tiles_per_wg; ++t) {
        double2* dst = reinterpret_cast<double2*>(out + ((uint64_t)blockIdx.x * tiles_per_wg + t) * tile_doubles);
        const double2 v = make_double2((double)t, (double)threadIdx.x);
        for (uint64_t i = threadIdx.x; i < tile_doubles / 2; i += 512) dst[i] = v;
    }
}
int main() {
    const uint64_t tile_doubles = 131072;              // 1 MiB
    double* d; hipMalloc(&d, 24ull << 30);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {8, 16, 32, 64, 96, 128, 192, 256}) {
        const int tiles = (int)((16ull << 30) / (1ull << 20) / wgs);   // 16 GiB in all
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 100 << 10, 0, d, tile_doubles, 4); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 100 << 10, 0, d, tile_doubles, tiles); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)wgs * tiles * (1 << 20);
        printf("%3d workgroups (CUs) storing: %.2f TB/s in all, %.1f GB/s per CU\n", wgs, bytes / ms / 1e9, bytes / ms / 1e6 / wgs);
    }
    return 0;
}
