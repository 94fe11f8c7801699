// microbenchmark: what does it cost to launch a kernel that does nothing, as a function of its static LDS size and block size?
// (k_pairing_deep<false>: 256 blocks x 1024 threads x 143 KB of LDS, no work at cfg3 -- and 58 us in the kernel trace.)
//     hipcc --offload-arch=gfx950 -O3 tools/mb/lds_launch.hip -o tools/mb/lds_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int LDS, int T>
__global__ __launch_bounds__(T) void k_empty(const uint32_t *n, uint32_t *out) {
    __shared__ uint32_t s[LDS / 4];
    if (blockIdx.x >= *n) return;
    s[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x] = s[(threadIdx.x * 7) % (LDS / 4)];
}
template <int LDS, int T> void run(int blocks, const uint32_t *n, uint32_t *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL((k_empty<LDS, T>), dim3(blocks), dim3(T), 0, 0, n, out);
    (void)hipEventRecord(e0, 0);
    for (int k = 0; k < 20; k++) hipLaunchKernelGGL((k_empty<LDS, T>), dim3(blocks), dim3(T), 0, 0, n, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B, %4d threads, %5d blocks: %.1f us per launch (20 back to back)\n", LDS, T, blocks, ms * 1000 / 20);
}
int main() {
    uint32_t *n, *out; (void)hipMalloc(&n, 4); (void)hipMalloc(&out, 4 << 20); (void)hipMemset(n, 0, 4);
    run<1024, 256>(256, n, out); run<1024, 1024>(256, n, out);
    run<16384, 1024>(256, n, out); run<65536, 1024>(256, n, out); run<66560, 1024>(256, n, out); run<98304, 1024>(256, n, out);
    run<143440, 1024>(256, n, out); run<143440, 256>(256, n, out); run<143440, 1024>(64, n, out); run<143440, 1024>(1024, n, out);
    run<81920, 1024>(256, n, out); run<81920, 1024>(512, n, out); run<40960, 1024>(1024, n, out);
    return 0;
}
