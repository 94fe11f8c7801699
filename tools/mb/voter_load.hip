// microbenchmark: per item (wave) fetch 6 voters x (75 B packed bases + 150 B quals), voters adjacent in memory.
//   A: per voter two instructions, 38 lanes x (2 B + 4 B)           (the layout of consensus pass A)
//   B: 16 B per lane, all six voters in one instruction per array     (5 + 10 lanes per voter)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint16_t u16u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
struct __attribute__((aligned(1))) U4 { uint32_t x, y, z, w; };
__global__ void k_a(const uint8_t *seq, const uint8_t *qual, uint32_t n_items, uint32_t *out) {
    uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; int lane = threadIdx.x & 63;
    if (item >= n_items) return;
    uint32_t acc = 0, mx = 0;
    for (int v = 0; v < 6; v++) {
        uint64_t r = (uint64_t)item * 6 + v;
        if (lane < 38) { acc |= *(const u16u *)(seq + r * 75 + 2 * lane); uint32_t q = *(const u32u *)(qual + r * 150 + 4 * lane); mx = mx > q ? mx : q; }
    }
    if ((acc ^ mx) == 0x12345678u) out[item] = acc;
}
__global__ void k_b(const uint8_t *seq, const uint8_t *qual, uint32_t n_items, uint32_t *out) {
    uint32_t item = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; int lane = threadIdx.x & 63;
    if (item >= n_items) return;
    uint32_t acc = 0, mx = 0;
    // bases: 6 voters x 5 chunks of 16 B = 30 lanes; quals: 6 x 10 = 60 lanes
    if (lane < 30) { int v = lane / 5, c = lane - v * 5; uint64_t r = (uint64_t)item * 6 + v; const U4 *p = (const U4 *)(seq + r * 75 + 16 * c); U4 a = *p; acc = a.x | a.y | a.z | a.w; }
    if (lane < 60) { int v = lane / 10, c = lane - v * 10; uint64_t r = (uint64_t)item * 6 + v; const U4 *p = (const U4 *)(qual + r * 150 + 16 * c); U4 a = *p; mx = a.x ^ a.y ^ a.z ^ a.w; }
    if ((acc ^ mx) == 0x12345678u) out[item] = acc;
}
int main() {
    const uint32_t items = 3100000; const uint64_t reads = (uint64_t)items * 6;
    uint8_t *seq, *qual; uint32_t *out;
    hipMalloc(&seq, reads * 75 + 64); hipMalloc(&qual, reads * 150 + 64); hipMalloc(&out, items * 4);
    hipMemset(seq, 1, reads * 75 + 64); hipMemset(qual, 2, reads * 150 + 64);
    dim3 grid((items * 64 + 255) / 256), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_a, grid, block, 0, 0, seq, qual, items, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("A 38 lanes x (2+4 B) x 6 voters : %.3f ms\n", ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k_b, grid, block, 0, 0, seq, qual, items, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("B 16 B per lane, 2 instructions : %.3f ms\n", ms);
    }
    printf("bytes %.2f GB\n", reads * 225.0 / 1e9);
    return 0;
}
