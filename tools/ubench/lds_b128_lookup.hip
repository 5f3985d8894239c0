// Micro-benchmark: 16-byte table lookups from LDS at random entries, as the general JSD kernel does them (one {invc, logc} pair per
// word and pair).  Patterns:
//   0  512 entries x 4 copies (32 KiB), copy = lane & 3          - the layout of po_valu_tiles.hip
//   1  8 192 entries, one copy (128 KiB), ds_read_b128           - exponent bits folded into the index
//   2  512 entries x 16 copies (128 KiB), copy = lane & 15       - no two lanes of a 16-lane group share a 16-byte column
//   3  8 192 entries, two 64 KiB tables of 8-byte halves, ds_read2st64_b64
//   4  8 192 entries x 8 bytes, one copy (64 KiB), ds_read_b64   - for reference
//   5  2 048 entries x 4 copies (128 KiB), copy = lane & 3
//   6  16 384 entries x 8 bytes, one copy (128 KiB), ds_read_b64  - a two-word sum table T[x1] + T[x2] of the JSD table kernel
//   7  16 384 entries x 8 bytes addressed as the kernel would: a wave-uniform row term + a per-lane column term, both sums of two
//      small counts (binomial around 29, as the folded counts of 2 kb contigs at k = 4) times 128 and times 1
// LDS cycles per wave-lookup = wave cycles / lookups issued per CU while the wave ran; clock from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int PAT>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (uint32_t i = threadIdx.x; i < 131072 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // 16 entry numbers per lane, stepped by a cheap permutation (x5 + odd) after use: 3 short vector instructions per lookup,
    // so the vector ALU stays far from limiting (the first version of this file spent ~10 per lookup and measured itself)
    uint32_t st = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 12345u;
    uint32_t e[16];
    for (int i = 0; i < 16; ++i) { st = st * 1664525u + 1013904223u; e[i] = st >> 12; }
    constexpr uint32_t emask = (PAT == 0 || PAT == 2) ? 511u : PAT == 5 ? 2047u : PAT >= 6 ? 16383u : 8191u;
    constexpr uint32_t eshift = PAT == 0 || PAT == 5 ? 6 : PAT == 2 ? 8 : PAT == 1 ? 4 : 3;
    const uint32_t lbase = base + (PAT == 0 || PAT == 5 ? (lane & 3) * 16 : PAT == 2 ? (lane & 15) * 16 : 0);
    uint32_t acc0 = 0, acc1 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e[r] = (e[r] * 5u + 0x9E3779B1u) & emask;
            uint32_t addr = (e[r] << eshift) + lbase;
            if (PAT == 7) {
                // counts ~ sum of 6 random bits x 2 ... cheap stand-in for a binomial: popcount of 6-bit fields, scaled to mean ~29
                const uint32_t x = e[r];
                const uint32_t c1 = __builtin_popcount(x & 0x7Fu) * 4 + ((x >> 7) & 7), c2 = __builtin_popcount((x >> 3) & 0x7Fu) * 4 + ((x >> 10) & 7);
                const uint32_t rowt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c1 * 128 + c2));   // wave-uniform row term
                addr = base + ((rowt + c1 * 128 + c2) & 16383u) * 8;
            }
            uint4 v = make_uint4(0, 0, 0, 0);
            if (PAT == 3) asm volatile("ds_read2st64_b64 %0, %1 offset1:128" : "=v"(v) : "v"(addr));
            else if (PAT == 4 || PAT >= 6) { uint2 u; asm volatile("ds_read_b64 %0, %1" : "=v"(u) : "v"(addr)); v.x = u.x; v.y = u.y; }
            else asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
            asm volatile("s_waitcnt lgkmcnt(6)");
            acc0 ^= v.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned long long* o = out + (blockIdx.x * 16 + (threadIdx.x >> 6)) * 3;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = acc0 ^ acc1;
    }
}

template <int PAT> void run(const char* name, unsigned long long* d, int waves = 8) {
    const int iters = 4000, blocks = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(64 * waves), 131072, 0, d, 50);
    hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(64 * waves), 131072, 0, d, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h((size_t)blocks * 16 * 3);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ghz = 0; int cnt = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { const double c = (double)h[(b * 16 + w) * 3], r = (double)h[(b * 16 + w) * 3 + 1]; cyc += c; ghz += c / (r * 10.0); ++cnt; }
    cyc /= cnt; ghz /= cnt;
    printf("%-58s %6.2f LDS cycles per wave-lookup (%d waves per CU), clock %.2f GHz\n", name, cyc / ((double)iters * 16 * waves), waves, ghz);
}

int main() {
    unsigned long long* d; (void)hipMalloc(&d, 8 << 20);
    run<0>("512 entries x 4 copies, b128 (in the tree)", d);
    run<5>("2 048 entries x 4 copies, b128", d);
    run<1>("8 192 entries, one copy, b128", d);
    run<2>("512 entries x 16 copies, b128 (conflict-free)", d);
    run<3>("8 192 entries, two 8-byte tables, read2st64_b64", d);
    run<4>("8 192 entries x 8 bytes, b64", d);
    run<4>("8 192 entries x 8 bytes, b64", d, 16);
    run<6>("16 384 entries x 8 bytes, b64 (two-word sum table)", d);
    run<6>("16 384 entries x 8 bytes, b64 (two-word sum table)", d, 16);
    run<7>("the same, row term + binomial column terms", d, 16);
    return 0;
}
