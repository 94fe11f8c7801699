// microbenchmark, companion of valu_rate.hip: issue rate of the instructions the pairing kernels are made of -- 64-bit shifts, 64-bit compares,
// v_lshl_add_u64, v_cndmask, DPP moves, v_bcnt / v_ffbl, v_readlane, ds_bpermute -- at 1 and 8 waves per SIMD.
//     hipcc --offload-arch=gfx950 -O3 tools/mb/valu_rate64.hip -o tools/mb/valu_rate64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
template <int KIND>
__global__ __launch_bounds__(64) void k_rate(unsigned long long *cyc, uint32_t *sink, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b0 = (uint32_t)a0 ^ 1, b1 = (uint32_t)a1 ^ 2, b2 = (uint32_t)a2 ^ 3, b3 = (uint32_t)a3 ^ 4, b4 = (uint32_t)a4 ^ 5, b5 = (uint32_t)a5 ^ 6, b6 = (uint32_t)a6 ^ 7, b7 = (uint32_t)a7 ^ 8;
    const uint32_t c = (seed & 7) | 1;
    const uint64_t c64 = seed | 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define OP8_64(ins) asm volatile(ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7\n" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c))
#define OP8_32(ins, tail) asm volatile(ins " %0, %0, %8 " tail "\n" ins " %1, %1, %8 " tail "\n" ins " %2, %2, %8 " tail "\n" ins " %3, %3, %8 " tail "\n" ins " %4, %4, %8 " tail "\n" ins " %5, %5, %8 " tail "\n" ins " %6, %6, %8 " tail "\n" ins " %7, %7, %8 " tail "\n" \
                               : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c))
        if (KIND == 0) OP8_64("v_lshrrev_b64");
        if (KIND == 1) OP8_64("v_lshlrev_b64");
        if (KIND == 2)      // 64-bit compare (the result goes to vcc; eight independent compares)
            asm volatile("v_cmp_lt_u64 vcc, %0, %8\nv_cmp_lt_u64 vcc, %1, %8\nv_cmp_lt_u64 vcc, %2, %8\nv_cmp_lt_u64 vcc, %3, %8\nv_cmp_lt_u64 vcc, %4, %8\nv_cmp_lt_u64 vcc, %5, %8\nv_cmp_lt_u64 vcc, %6, %8\nv_cmp_lt_u64 vcc, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c64) : "vcc");
        if (KIND == 3)
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cmp_lt_u32 vcc, %1, %8\nv_cmp_lt_u32 vcc, %2, %8\nv_cmp_lt_u32 vcc, %3, %8\nv_cmp_lt_u32 vcc, %4, %8\nv_cmp_lt_u32 vcc, %5, %8\nv_cmp_lt_u32 vcc, %6, %8\nv_cmp_lt_u32 vcc, %7, %8\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c) : "vcc");
        if (KIND == 4)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\nv_lshl_add_u64 %4, %4, 0, %8\nv_lshl_add_u64 %5, %5, 0, %8\nv_lshl_add_u64 %6, %6, 0, %8\nv_lshl_add_u64 %7, %7, 0, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c64));
        if (KIND == 5) OP8_32("v_cndmask_b32", ", vcc");
        if (KIND == 6)
            asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        if (KIND == 7) OP8_32("v_bcnt_u32_b32", "");
        if (KIND == 8)
            asm volatile("v_ffbl_b32 %0, %0\nv_ffbl_b32 %1, %1\nv_ffbl_b32 %2, %2\nv_ffbl_b32 %3, %3\nv_ffbl_b32 %4, %4\nv_ffbl_b32 %5, %5\nv_ffbl_b32 %6, %6\nv_ffbl_b32 %7, %7\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        if (KIND == 9)      // ds_bpermute: eight independent ones, then the wait
            asm volatile("ds_bpermute_b32 %0, %8, %0\nds_bpermute_b32 %1, %8, %1\nds_bpermute_b32 %2, %8, %2\nds_bpermute_b32 %3, %8, %3\nds_bpermute_b32 %4, %8, %4\nds_bpermute_b32 %5, %8, %5\nds_bpermute_b32 %6, %8, %6\nds_bpermute_b32 %7, %8, %7\ns_waitcnt lgkmcnt(0)\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
        if (KIND == 10) OP8_32("v_and_b32", "");
        if (KIND == 11) OP8_32("v_or_b32", "");
        if (KIND == 12) OP8_32("v_lshrrev_b32", "");
        if (KIND == 13)     // v_cmp into an SGPR pair + v_cndmask on it: the select idiom
            asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_u32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_u32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c) : "vcc");
        if (KIND == 15) {    // v_cndmask on an SGPR pair nobody writes inside the loop
            asm volatile("s_mov_b64 s[20:21], 0x55\n" ::: "s20", "s21");
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\nv_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c) : "s20", "s21");
        }
        if (KIND == 16)      // SALU writes the mask, the VALU select reads it (the compiler's s_and_b64 + v_cndmask_b32_e64 idiom): 4 + 4
            asm volatile("s_and_b64 s[20:21], exec, s[22:23]\ns_nop 1\nv_cndmask_b32_e64 %0, %0, %4, s[20:21]\ns_and_b64 s[20:21], exec, s[22:23]\ns_nop 1\nv_cndmask_b32_e64 %1, %1, %4, s[20:21]\ns_and_b64 s[20:21], exec, s[22:23]\ns_nop 1\nv_cndmask_b32_e64 %2, %2, %4, s[20:21]\ns_and_b64 s[20:21], exec, s[22:23]\ns_nop 1\nv_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(c) : "s20", "s21", "s22", "s23", "scc");
        if (KIND == 17) OP8_32("v_max_u32", "");
        if (KIND == 18) OP8_32("v_add_u32", "");
        if (KIND == 19) OP8_32("v_lshlrev_b32", "");
        if (KIND == 20)      // the same and with sixteen chains in flight (valu_rate.hip's shape)
            asm volatile("v_and_b32 %0, %0, %8\nv_and_b32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_and_b32 %3, %3, %8\nv_and_b32 %4, %4, %8\nv_and_b32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_and_b32 %7, %7, %8\n"
                         "v_max_u32 %0, %0, %8\nv_max_u32 %1, %1, %8\nv_max_u32 %2, %2, %8\nv_max_u32 %3, %3, %8\nv_max_u32 %4, %4, %8\nv_max_u32 %5, %5, %8\nv_max_u32 %6, %6, %8\nv_max_u32 %7, %7, %8\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
        if (KIND == 14)
            asm volatile("v_readlane_b32 s20, %0, 3\nv_readlane_b32 s21, %1, 3\nv_readlane_b32 s22, %2, 3\nv_readlane_b32 s23, %3, 3\nv_readlane_b32 s24, %4, 3\nv_readlane_b32 s25, %5, 3\nv_readlane_b32 s26, %6, 3\nv_readlane_b32 s27, %7, 3\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    const uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    if (r == 0x12345678u) *sink = (uint32_t)r;
}
template <int KIND> void run(const char *name, int per_iter, unsigned long long *cyc, uint32_t *sink, int n_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 8}) {
        const int blocks = n_cu * 4 * wps;
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, sink, 7u);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, sink, 7u);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)ITER * per_iter;
        printf("%-22s waves/SIMD %d: kernel %.3f ms -> %.3f wave instructions per SIMD per ns\n", name, wps, ms, instr * blocks / ms / 1e6 / (n_cu * 4));
    }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    unsigned long long *cyc; uint32_t *sink; hipMalloc(&cyc, 8 * 65536); hipMalloc(&sink, 4);
    const int n = pr.multiProcessorCount;
    run<0>("v_lshrrev_b64", 8, cyc, sink, n);
    run<1>("v_lshlrev_b64", 8, cyc, sink, n);
    run<2>("v_cmp_lt_u64", 8, cyc, sink, n);
    run<3>("v_cmp_lt_u32", 8, cyc, sink, n);
    run<4>("v_lshl_add_u64", 8, cyc, sink, n);
    run<5>("v_cndmask_b32", 8, cyc, sink, n);
    run<6>("v_mov_b32_dpp", 8, cyc, sink, n);
    run<7>("v_bcnt_u32_b32", 8, cyc, sink, n);
    run<8>("v_ffbl_b32", 8, cyc, sink, n);
    run<9>("ds_bpermute_b32", 8, cyc, sink, n);
    run<10>("v_and_b32", 8, cyc, sink, n);
    run<11>("v_or_b32", 8, cyc, sink, n);
    run<12>("v_lshrrev_b32", 8, cyc, sink, n);
    run<13>("v_cmp+v_cndmask", 8, cyc, sink, n);
    run<14>("v_readlane_b32", 8, cyc, sink, n);
    run<15>("v_cndmask_e64 s[..]", 8, cyc, sink, n);
    run<16>("s_and + nop + cndmask", 8, cyc, sink, n);
    run<17>("v_max_u32", 8, cyc, sink, n);
    run<18>("v_add_u32", 8, cyc, sink, n);
    run<19>("v_lshlrev_b32", 8, cyc, sink, n);
    run<20>("v_and + v_max mixed", 16, cyc, sink, n);
    return 0;
}
