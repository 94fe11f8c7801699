// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS PATTERNS of this engine (MI355X_MICROARCH.md: "FETCH_SIZE
// reports 1/2 of the bytes of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel reads (or writes) a buffer of KNOWN size exactly once, far beyond the
// 256 MB Infinity Cache; run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/fetch_calib.sh) and divide.
//   c16 / c8 / c4 / c2 / c1   coalesced loads of 16 / 8 / 4 / 2 / 1 bytes per lane
//   vote_a                    k_vote pass A: lane = (read, 16-column chunk): one unaligned 8-byte load of packed bases, two of qualities,
//                             reads of 75 + 150 bytes back to back (the benchmark's layout)
//   desc32                    k_vote P1 / k_out_gather: one 32-byte record (two 16-byte loads) per lane from every 8th record (scattered sectors)
//   w16 / w4                  coalesced stores of 16 / 4 bytes per lane;  w16s: 16-byte stores to every 4th 64-byte line (scattered)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64u __attribute__((aligned(1)));
template <class T> __global__ void k_coalesced(const T *p, uint64_t n, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; T v = p[i]; uint32_t x = 0; const uint8_t *b = (const uint8_t *)&v; for (unsigned k = 0; k < sizeof(T); k++) x += b[k]; if (x == (uint32_t)i * 2654435761u + 77u) *sink = x; }
__global__ void k_c16(const uint4 *p, uint64_t n, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const uint4 v = p[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = v.x; }
__global__ void k_vote_a(const uint8_t *seq, const uint8_t *qual, uint64_t n_reads, uint32_t *sink) {
    const uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, r = it / 10, c = it % 10;
    if (r >= n_reads) return;
    const uint64_t s = *(const u64u *)(seq + r * 75 + 8 * c), q0 = *(const u64u *)(qual + r * 150 + 16 * c), q1 = *(const u64u *)(qual + r * 150 + 16 * c + 8);
    if ((s ^ q0 ^ q1) == 0x123456789ull) *sink = (uint32_t)s;
}
__global__ void k_desc32(const uint4 *p, uint64_t n_rec, uint32_t *sink) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n_rec) return; const uint4 a = p[16 * i], b = p[16 * i + 1]; if ((a.x ^ b.y) == 0x12345u) *sink = a.x; }
__global__ void k_w16(uint4 *p, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4((uint32_t)i, 1, 2, 3); }
__global__ void k_w4(uint32_t *p, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (uint32_t)i; }
__global__ void k_w16s(uint4 *p, uint64_t n) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[16 * i] = make_uint4((uint32_t)i, 1, 2, 3); }
int main() {
    const uint64_t bytes = 2ull << 30;                       // 2 GiB per array
    uint8_t *a, *b; uint32_t *sink;
    hipMalloc(&a, bytes + 64); hipMalloc(&b, bytes + 64); hipMalloc(&sink, 8);
    hipMemset(a, 1, bytes + 64); hipMemset(b, 2, bytes + 64); hipDeviceSynchronize();
    auto g = [](uint64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    const uint64_t n_reads = bytes / 150;                    // qual array full; seq array half used
    hipLaunchKernelGGL(k_c16, g(bytes / 16), dim3(256), 0, 0, (const uint4 *)a, bytes / 16, sink);
    hipLaunchKernelGGL(k_coalesced<uint64_t>, g(bytes / 8), dim3(256), 0, 0, (const uint64_t *)a, bytes / 8, sink);
    hipLaunchKernelGGL(k_coalesced<uint32_t>, g(bytes / 4), dim3(256), 0, 0, (const uint32_t *)a, bytes / 4, sink);
    hipLaunchKernelGGL(k_coalesced<uint16_t>, g(bytes / 2), dim3(256), 0, 0, (const uint16_t *)a, bytes / 2, sink);
    hipLaunchKernelGGL(k_coalesced<uint8_t>, g(bytes / 2), dim3(256), 0, 0, (const uint8_t *)a, bytes / 2, sink);      // (1 GiB: a byte per lane is slow)
    hipLaunchKernelGGL(k_vote_a, g(n_reads * 10), dim3(256), 0, 0, (const uint8_t *)a, (const uint8_t *)b, n_reads, sink);
    hipLaunchKernelGGL(k_desc32, g(bytes / 256), dim3(256), 0, 0, (const uint4 *)a, bytes / 256, sink);
    hipLaunchKernelGGL(k_w16, g(bytes / 16), dim3(256), 0, 0, (uint4 *)b, bytes / 16);
    hipLaunchKernelGGL(k_w4, g(bytes / 4), dim3(256), 0, 0, (uint32_t *)b, bytes / 4);
    hipLaunchKernelGGL(k_w16s, g(bytes / 256), dim3(256), 0, 0, (uint4 *)b, bytes / 256);
    hipDeviceSynchronize();
    printf("known bytes: c16 c8 c4 c2 %llu; c1 %llu; vote_a %llu (useful) ; desc32 %llu useful in %llu sectors of 64 B; w16 w4 %llu; w16s %llu useful\n",
           (unsigned long long)bytes, (unsigned long long)bytes / 2, (unsigned long long)(n_reads * 225), (unsigned long long)(bytes / 256 * 32), (unsigned long long)(bytes / 256), (unsigned long long)bytes, (unsigned long long)(bytes / 256 * 16));
    return 0;
}
