// Device-to-host paths for a large result (row blocks of the distance matrix): what each costs on this box.
//   1 hipMemcpy into fresh pageable memory             2 hipHostMalloc + hipMemcpy into it (alloc timed apart)
//   3 hipHostRegister of fresh pageable memory + copy  4 pinned ring (2 x 32 MB) + T host threads memcpy, plain pages
//   5 the same with MADV_HUGEPAGE on the destination   6 pinned ring of 4 x 64 MB on two streams
// build: hipcc -O3 --offload-arch=gfx950 d2h_paths.hip -o d2h_paths -lpthread ; run: ./d2h_paths [GB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err__)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void* fresh(size_t bytes, bool huge) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    if (huge) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

static void ring_copy(const char* d_src, char* dst, size_t bytes, int nbuf, size_t chunk, int nthr, int nstreams) {
    std::vector<void*> stage(nbuf);
    std::vector<hipStream_t> st(nstreams);
    std::vector<hipEvent_t> ev(nbuf);
    for (auto& s : stage) CK(hipHostMalloc(&s, chunk, hipHostMallocDefault));
    for (auto& s : st) CK(hipStreamCreate(&s));
    for (auto& e : ev) CK(hipEventCreate(&e));
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    auto issue = [&](size_t c) {
        const size_t o = c * chunk, len = std::min(chunk, bytes - o);
        CK(hipMemcpyAsync(stage[c % nbuf], d_src + o, len, hipMemcpyDeviceToHost, st[c % nstreams]));
        CK(hipEventRecord(ev[c % nbuf], st[c % nstreams]));
    };
    for (size_t c = 0; c < (size_t)nbuf - 1 && c < nchunks; ++c) issue(c);
    for (size_t c = 0; c < nchunks; ++c) {
        if (c + nbuf - 1 < nchunks) issue(c + nbuf - 1);
        CK(hipEventSynchronize(ev[c % nbuf]));
        const size_t o = c * chunk, len = std::min(chunk, bytes - o);
        const char* src = static_cast<const char*>(stage[c % nbuf]);
        std::vector<std::thread> th;
        const size_t per = (len + nthr - 1) / nthr;
        for (int t = 0; t < nthr; ++t)
            th.emplace_back([=]() { const size_t a = t * per; if (a < len) memcpy(dst + o + a, src + a, std::min(per, len - a)); });
        for (auto& x : th) x.join();
    }
    for (auto& s : stage) CK(hipHostFree(s));
    for (auto& s : st) CK(hipStreamDestroy(s));
    for (auto& e : ev) CK(hipEventDestroy(e));
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t bytes = (size_t)(gb * 1e9) & ~(size_t)4095;
    char* d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    CK(hipDeviceSynchronize());
    const unsigned hw = std::thread::hardware_concurrency();
    printf("%.2f GB, %u host threads\n", bytes / 1e9, hw);
    double t;
    { void* h = fresh(bytes, false); t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("1 hipMemcpy -> fresh pageable             %7.1f ms  %5.1f GB/s\n", t * 1e3, bytes / t / 1e9); munmap(h, bytes); }
    { void* h; t = now(); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); double ta = now() - t;
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("2 hipHostMalloc %7.1f ms, copy            %7.1f ms  %5.1f GB/s (copy only)\n", ta * 1e3, t * 1e3, bytes / t / 1e9);
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      printf("  second copy into the same pinned buffer   %7.1f ms  %5.1f GB/s\n", t * 1e3, bytes / t / 1e9);
      t = now(); CK(hipHostFree(h)); printf("  hipHostFree %7.1f ms\n", (now() - t) * 1e3); }
    for (int huge = 0; huge < 2; ++huge) {
      void* h = fresh(bytes, huge); t = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double tr = now() - t;
      t = now(); CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); t = now() - t;
      double tu = now(); CK(hipHostUnregister(h)); tu = now() - tu;
      printf("3 hipHostRegister(fresh%s) %7.1f ms, copy %7.1f ms  %5.1f GB/s, unregister %.1f ms -> all %5.1f GB/s\n", huge ? ", THP" : "", tr * 1e3, t * 1e3,
             bytes / t / 1e9, tu * 1e3, bytes / (tr + t + tu) / 1e9); munmap(h, bytes); }
    for (int huge = 0; huge < 2; ++huge)
        for (int nthr : {8, 16, 32}) {
            if ((unsigned)nthr > hw && nthr != 8) continue;
            void* h = fresh(bytes, huge); t = now(); ring_copy(d, (char*)h, bytes, 2, 32u << 20, nthr, 1); t = now() - t;
            printf("%d ring 2 x 32 MB, %2d threads, %s        %7.1f ms  %5.1f GB/s\n", huge ? 5 : 4, nthr, huge ? "THP  " : "plain", t * 1e3, bytes / t / 1e9);
            munmap(h, bytes);
        }
    for (int nthr : {8, 16}) {
        void* h = fresh(bytes, true); t = now(); ring_copy(d, (char*)h, bytes, 4, 64u << 20, nthr, 2); t = now() - t;
        printf("6 ring 4 x 64 MB, 2 streams, %2d threads, THP %7.1f ms  %5.1f GB/s\n", nthr, t * 1e3, bytes / t / 1e9); munmap(h, bytes);
    }
    { void* h = fresh(bytes, true); t = now(); memset(h, 0, bytes); t = now() - t;
      printf("  (first touch of %.1f GB with THP by one thread: %.1f ms)\n", bytes / 1e9, t * 1e3); munmap(h, bytes); }
    CK(hipFree(d));
    return 0;
}
