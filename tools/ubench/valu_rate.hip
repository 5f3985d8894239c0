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
            if (OP == 8) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            if (OP == 9) asm volatile("v_bfe_u32 %0, %0, 13, 7" : "+v"(x));
            if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x) : "v"(c));
            if (OP == 11) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(c));
            if (OP == 13) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(x));
            if (OP == 14) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
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
template <int OP>
__global__ void kd(double* out, int iters) {
    double a[8]; for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
    double c = out[0] + 1.000001, d = 0.999999;
    unsigned u = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            double& x = a[r & 7];
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            if (OP == 3) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x) : "v"(u));
            if (OP == 4) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
            if (OP == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
            if (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(c));       // 64-bit integer add (fixed-point accumulate?)
            if (OP == 7) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = s;
}
template <int OP> void rund(const char* name, double* d, int waves_per_simd) {
    const int iters = 2048; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 4 * waves_per_simd / 4), block(256);
    hipLaunchKernelGGL(kd<OP>, grid, block, 0, 0, d, 16); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kd<OP>, grid, block, 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s waves/SIMD=%d: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / ((double)iters * REP * waves_per_simd));
}
int main() {
    unsigned* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_min_i16", d, w); run<1>("v_min_i32", d, w); run<2>("v_pk_sub_i16", d, w); run<3>("v_perm_b32", d, w);
        run<4>("v_and_b32", d, w); run<5>("v_sub_u32", d, w); run<6>("v_bitop3_b32", d, w); run<7>("v_sad_u8", d, w);
        run<8>("v_and_or_b32", d, w); run<9>("v_bfe_u32", d, w); run<10>("v_lshl_or_b32", d, w); run<11>("v_add3_u32", d, w);
        run<12>("v_mov_b32", d, w); run<13>("v_lshrrev_b32", d, w); run<14>("v_med3_i32", d, w);
        double* dd = reinterpret_cast<double*>(d);
        rund<0>("v_fma_f64", dd, w); rund<1>("v_add_f64", dd, w); rund<2>("v_mul_f64", dd, w); rund<3>("v_cvt_f64_u32", dd, w);
        rund<4>("v_frexp_mant_f64", dd, w); rund<5>("v_rsq_f64", dd, w); rund<6>("v_lshl_add_u64", dd, w); rund<7>("v_fmac_f64", dd, w);
    }
    return 0;
}
