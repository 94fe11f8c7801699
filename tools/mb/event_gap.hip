// microbenchmark: what does a timing event between two kernels cost?  A chain of 64 short kernels (a) back to back, (b) with hipEventRecord behind each,
// (c) each launched through hipExtLaunchKernelGGL with a stop event attached to the dispatch itself.
//     hipcc --offload-arch=gfx950 -O3 tools/mb/event_gap.hip -o tools/mb/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
__global__ __launch_bounds__(256) void k_short(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] * 3u + 1u;
}
int main() {
    const uint32_t n = 1u << 22; uint32_t *p; (void)hipMalloc(&p, n * 4); (void)hipMemset(p, 1, n * 4);
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t ev[65]; for (auto &e : ev) (void)hipEventCreate(&e);
    hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    for (int mode = 0; mode < 3; mode++) for (int rep = 0; rep < 3; rep++) {
        (void)hipStreamSynchronize(s);
        const auto h0 = std::chrono::steady_clock::now();
        (void)hipEventRecord(t0, s);
        for (int k = 0; k < 64; k++) {
            if (mode == 2) hipExtLaunchKernelGGL(k_short, dim3(n / 256), dim3(256), 0, s, nullptr, ev[k], 0, p, n);
            else hipLaunchKernelGGL(k_short, dim3(n / 256), dim3(256), 0, s, p, n);
            if (mode == 1) (void)hipEventRecord(ev[k], s);
        }
        (void)hipEventRecord(t1, s); (void)hipEventSynchronize(t1);
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        float ms = 0, last = 0; (void)hipEventElapsedTime(&ms, t0, t1);
        if (mode) (void)hipEventElapsedTime(&last, ev[62], ev[63]);
        printf("%-44s 64 kernels: %.1f us on the stream (%.2f us per kernel), host %.0f us; event 62 -> 63: %.2f us\n",
               mode == 0 ? "back to back" : mode == 1 ? "hipEventRecord behind every kernel" : "hipExtLaunchKernelGGL with a stop event", ms * 1000, ms * 1000 / 64, host_us, last * 1000);
    }
    return 0;
}
