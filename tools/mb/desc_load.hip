// microbenchmark: three 16-byte loads per lane from one 48-byte record (same cache line, separate instructions)
// versus three lanes loading the record in ONE instruction + ds_bpermute redistribution.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct __attribute__((aligned(16))) Rec { uint4 q[3]; };
__global__ void k_separate(const Rec *r, const uint32_t *idx, uint32_t n, uint32_t *out, int lanes_used) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t item = g >> 6; int lane = threadIdx.x & 63;
    if (item * lanes_used >= n) return;
    uint32_t acc = 0;
    if (lane < lanes_used) {
        uint32_t i = idx[item * lanes_used + lane];
        const uint4 *p = (const uint4 *)(r + i);
        uint4 a = p[0], b = p[1], c = p[2];
        acc = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w;
    }
    if (acc == 0x12345678u) out[g] = acc;
}
__global__ void k_coop(const Rec *r, const uint32_t *idx, uint32_t n, uint32_t *out, int lanes_used) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t item = g >> 6; int lane = threadIdx.x & 63;
    if (item * lanes_used >= n) return;
    // lane l loads chunk (l % 3) of record (l / 3): one instruction covers up to 21 records
    uint32_t acc = 0;
    int rec = lane / 3, ch = lane - rec * 3;
    uint32_t i = rec < lanes_used ? idx[item * lanes_used + rec] : 0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rec < lanes_used) v = ((const uint4 *)(r + i))[ch];
    // gather the three chunks of record `lane` from lanes 3*lane .. 3*lane+2
    uint32_t x = 0;
    for (int c = 0; c < 3; c++) {
        int src = (3 * lane + c) << 2;
        x ^= __builtin_amdgcn_ds_bpermute(src, v.x) ^ __builtin_amdgcn_ds_bpermute(src, v.y) ^ __builtin_amdgcn_ds_bpermute(src, v.z) ^ __builtin_amdgcn_ds_bpermute(src, v.w);
    }
    if (lane < lanes_used) acc = x;
    if (acc == 0x12345678u) out[g] = acc;
}
int main() {
    const uint32_t N = 20000000; const int LU = 6;
    Rec *r; uint32_t *idx, *out;
    hipMalloc(&r, (size_t)N * sizeof(Rec)); hipMalloc(&idx, (size_t)N * 4); hipMalloc(&out, (size_t)(N / LU + 1) * 64 * 4);
    hipMemset(r, 1, (size_t)N * sizeof(Rec));
    std::vector<uint32_t> h(N);
    for (uint32_t i = 0; i < N; i++) h[i] = (i / 12) * 12 + ((i * 7) % 12);      // locally shuffled: neighbours stay within 12 records
    hipMemcpy(idx, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    uint32_t items = N / LU; dim3 grid((items * 64 + 255) / 256), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_separate, grid, block, 0, 0, r, idx, N, out, LU); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("separate 3x16B per lane : %.3f ms\n", ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k_coop, grid, block, 0, 0, r, idx, N, out, LU); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("cooperative 1 instr     : %.3f ms\n", ms);
    }
    return 0;
}
