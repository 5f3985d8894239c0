// How fast can P processes put disjoint column ranges of every row of one float32 container into the page cache?
// (the multi-GPU form of --large memmap: every rank owns some columns of some rows)
//   mode pwrite : one pwrite(2) per row piece, T threads per process (buffered writes take the inode lock: they serialise,
//                 and with small pieces from several processes the lock changes hands at every call)
//   mode mmap   : the file mapped MAP_SHARED, T threads per process memcpy their row pieces (page faults, no inode lock)
//   mode mmap+pop: the same after madvise(MADV_POPULATE_WRITE) of the process's row range (faults taken in bulk)
//   mode big    : each process writes a contiguous half of the file with 64 MiB pwrites (what row-complete slabs allow)
// usage: container_write_modes <path> <n> <procs> <threads> <mode>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage\n"); return 2; }
    const char* path = argv[1];
    const uint64_t n = strtoull(argv[2], nullptr, 10);
    const int P = atoi(argv[3]), T = atoi(argv[4]);
    const std::string mode = argv[5];
    const uint64_t row = n * 4, total = n * row;
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (fd < 0 || fallocate(fd, 0, 0, (off_t)total) != 0) { if (fd < 0 || ftruncate(fd, (off_t)total)) { perror("file"); return 1; } }
    close(fd);
    const uint64_t piece = row / P;                         // process p owns columns [p * piece, (p + 1) * piece) of every row
    std::vector<float> src(piece / 4 * 1024);               // 1024 source rows, reused
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)i;
    const double t0 = now();
    std::vector<pid_t> kids;
    for (int p = 0; p < P; ++p) {
        pid_t k = fork();
        if (k == 0) {
            int f = open(path, O_RDWR);
            uint8_t* map = nullptr;
            if (mode.rfind("mmap", 0) == 0) {
                map = (uint8_t*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, f, 0);
                if (map == MAP_FAILED) { perror("mmap"); _exit(1); }
            }
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t]() {
                    const uint64_t per = (n + T - 1) / T, r0 = std::min<uint64_t>(n, t * per), r1 = std::min<uint64_t>(n, r0 + per);
                    if (mode == "big") {
                        const uint64_t half = total / P, a = p * half + (half / T) * t, b = (t + 1 == T) ? (p + 1) * half : a + half / T;
                        for (uint64_t at = a; at < b;) {
                            const uint64_t len = std::min<uint64_t>(b - at, 64u << 20);
                            ssize_t w = pwrite(f, (const uint8_t*)src.data() + (at % (src.size() * 2)), std::min<uint64_t>(len, src.size() * 2), (off_t)at);
                            if (w <= 0) { perror("pwrite"); _exit(1); }
                            at += (uint64_t)w;
                        }
                        return;
                    }
                    if (mode == "mmap+pop" && r1 > r0) {
                        const uint64_t a = (r0 * row) & ~4095ull, b = std::min<uint64_t>(total, (r1 * row + 4095) & ~4095ull);
                        if (madvise(map + a, b - a, MADV_POPULATE_WRITE) != 0) perror("madvise");
                    }
                    for (uint64_t r = r0; r < r1; ++r) {
                        const uint8_t* s = (const uint8_t*)src.data() + (r % 1024) * piece;
                        const uint64_t at = r * row + p * piece;
                        if (map) memcpy(map + at, s, piece);
                        else {
                            uint64_t done = 0;
                            while (done < piece) { ssize_t w = pwrite(f, s + done, piece - done, (off_t)(at + done)); if (w <= 0) _exit(1); done += (uint64_t)w; }
                        }
                    }
                });
            for (auto& x : th) x.join();
            if (map) munmap(map, total);
            close(f);
            _exit(0);
        }
        kids.push_back(k);
    }
    int bad = 0;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); bad |= st; }
    const double dt = now() - t0;
    printf("%-9s procs %d threads %d  n %llu  %.2f GB in %.3f s = %.2f GB/s%s\n", mode.c_str(), P, T, (unsigned long long)n, total / 1e9, dt,
           total / 1e9 / dt, bad ? "  (a child FAILED)" : "");
    unlink(path);
    return bad ? 1 : 0;
}
