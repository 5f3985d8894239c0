// Micro-benchmark: issue rate of a few VALU instructions on gfx950 (cycles per wave-instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP>
__global__ void k(unsigned* out, int iters) {
    unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    unsigned c = out[0], d = 0x00010001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            unsigned& x = a[r & 7];
            if (OP == 0) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(d));
            if (OP == 1) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 2) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "v"(d));
            if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            if (OP == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 5) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 6) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36" : "+v"(x) : "v"(c), "v"(d));
            if (OP == 7) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = s;
}
template <int OP> void run(const char* name, unsigned* d, int waves_per_simd) {
    const int iters = 4096; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 4 * waves_per_simd / 4), block(256);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 16); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_simd = (double)iters * REP * waves_per_simd;
    printf("%-14s waves/SIMD=%d: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
    unsigned* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_min_i16", d, w); run<1>("v_min_i32", d, w); run<2>("v_pk_sub_i16", d, w); run<3>("v_perm_b32", d, w);
        run<4>("v_and_b32", d, w); run<5>("v_sub_u32", d, w); run<6>("v_bitop3_b32", d, w); run<7>("v_sad_u8", d, w);
    }
    return 0;
}
