// what a device allocation costs on this platform: hipMalloc / first touch / hipFree of buffers of 64 MB .. 8 GB, fresh and repeated, one and four host threads
// (the question behind the sharded file runner's 0.4 - 1.9 s on some boxes): hipcc --offload-arch=gfx950 -O2 tools/mb/alloc_cost.hip -o tools/mb/alloc_cost -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; rep++)
        for (size_t mb : {64, 512, 2048, 8192}) {
            void *p = nullptr; const size_t n = mb << 20;
            double t0 = now(); if (hipMalloc(&p, n) != hipSuccess) { printf("hipMalloc %zu MB failed\n", mb); continue; }
            double t1 = now(); (void)hipMemset(p, 0, n); (void)hipDeviceSynchronize();
            double t2 = now(); (void)hipMemset(p, 1, n); (void)hipDeviceSynchronize();
            double t3 = now(); (void)hipFree(p);
            double t4 = now();
            printf("rep %d  %5zu MB: hipMalloc %.4f s (%.3f s/GB)  first memset %.4f s  second memset %.4f s  hipFree %.4f s\n", rep, mb, t1 - t0, (t1 - t0) / (mb / 1024.0), t2 - t1, t3 - t2, t4 - t3);
        }
    for (int nt : {1, 4}) {
        std::vector<std::thread> th; std::vector<void *> ps(nt, nullptr);
        double t0 = now();
        for (int k = 0; k < nt; k++) th.emplace_back([&, k] { for (int q = 0; q < 40; q++) { void *p = nullptr; (void)hipMalloc(&p, (size_t)100 << 20); if (q == 0) ps[k] = p; } });
        for (auto &t : th) t.join();
        printf("%d thread(s) x 40 allocations of 100 MB: %.4f s\n", nt, now() - t0);
    }
    return 0;
}
