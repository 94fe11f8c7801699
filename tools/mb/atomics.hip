// microbenchmark: what bounds the leader kernel of the cluster formation (gce_cluster.hpp: one CAS + one 64-bit add per (cluster, block))?
// n ops over a table of 16-byte entries, one op per lane, address pattern "sliding" (entry = 6 * i + hash(i) % 6: neighbouring lanes touch
// neighbouring lines, the table is 6 x n entries) or "random".  Variants: device-scope CAS, add with / without return, CAS + dependent add
// (the leader pattern), relaxed load + CAS (one read-modify-write instead of two), workgroup-scope atomics (execute in the XCD's L2:
// NOT coherent across XCDs, shown for the cost only), plain 16-byte loads and stores as the memory-system reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((aligned(16))) Ent { unsigned long long key, ic; };
__device__ __forceinline__ uint64_t addr_of(uint64_t i, uint64_t n, int random) {
    if (!random) { uint32_t h = (uint32_t)i * 0x9E3779B1u; return 6 * i + (h >> 29) % 6; }
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    return x % (6 * n);
}
template <int MODE>
__global__ void k_op(Ent *tab, uint64_t n, int random, unsigned long long *sink) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ent *e = tab + addr_of(i, n, random);
    unsigned long long r = 0;
    if (MODE == 0) r = atomicCAS(&e->key, 0ull, i + 1);                                                   // device CAS
    if (MODE == 1) r = atomicAdd(&e->ic, 3ull);                                                           // device add, returning
    if (MODE == 2) { atomicAdd(&e->ic, 3ull); }                                                           // device add, no return
    if (MODE == 3) { r = atomicCAS(&e->key, 0ull, i + 1); r += atomicAdd(&e->ic, 3ull + (r & 1)); }       // CAS, then dependent add (leader pattern)
    if (MODE == 4) { unsigned long long c = __hip_atomic_load(&e->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // load + one CAS carrying the count
                     for (;;) { const unsigned long long want = c ? c + 3 : ((i + 1) << 24) + 3; const unsigned long long o = atomicCAS(&e->key, c, want); if (o == c) break; c = o; } r = c; }
    if (MODE == 5) r = __hip_atomic_compare_exchange_strong(&e->key, &r, i + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0 : r;   // L2-scope CAS
    if (MODE == 6) { unsigned long long z = 0; __hip_atomic_compare_exchange_strong(&e->key, &z, i + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                     r = z + __hip_atomic_fetch_add(&e->ic, 3ull + (z & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // L2-scope CAS + add
    if (MODE == 7) { const uint4 v = *reinterpret_cast<const uint4 *>(e); r = v.x ^ v.z; }                 // plain 16-byte load
    if (MODE == 8) { *reinterpret_cast<uint4 *>(e) = make_uint4((uint32_t)i, 0, 1, 0); }                   // plain 16-byte store
    if (MODE == 9) { r = atomicCAS((unsigned int *)&e->key, 0u, (unsigned int)i + 1u); }                   // 32-bit device CAS
    if (MODE == 10) { r = atomicCAS(&e->key, 0ull, i + 1); const uint4 v = *reinterpret_cast<const uint4 *>(tab + addr_of(i ^ 1, n, random)); r += v.x; }   // CAS + independent load
    if (r == 0x123456789abcull) *sink = r;
}
int main() {
    const uint64_t n = 7000000;
    Ent *tab; unsigned long long *sink;
    hipMalloc(&tab, 6 * n * sizeof(Ent)); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *nm[11] = {"device CAS u64", "device add u64 (return)", "device add u64 (no return)", "device CAS -> dependent add", "load + one CAS (count in the word)",
                          "workgroup-scope CAS u64", "workgroup-scope CAS -> add", "plain 16 B load", "plain 16 B store", "device CAS u32", "device CAS + independent load"};
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (int random = 0; random < 2; random++)
        for (int mode = 0; mode < 11; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(tab, 0, 6 * n * sizeof(Ent)); hipDeviceSynchronize();
                hipEventRecord(e0);
                switch (mode) {
#define C(M) case M: hipLaunchKernelGGL(k_op<M>, grid, block, 0, 0, tab, n, random, sink); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10)
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%-8s %-36s %.3f ms  %.1f G lanes/s\n", random ? "random" : "sliding", nm[mode], best, n / best / 1e6);
        }
    return 0;
}
