// microbenchmark: how fast does a gfx950 SIMD issue plain integer VALU instructions of wave64 waves?  (Is k_vote, at 1751 VALU instructions
// per wave and seven waves per SIMD, near its issue bound or far from it?)  Every wave runs ITER x 16 independent instructions of one
// kind on 16 accumulators (inline asm, no memory); waves per SIMD = 1, 2, 4, 7, 8.  Printed: cycles (s_memtime, shader clock) per
// instruction of one wave, and instructions per cycle per SIMD.      hipcc --offload-arch=gfx950 -O3 tools/mb/valu_rate.hip -o tools/mb/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(64) void k_rate(unsigned long long *cyc, uint32_t *sink, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b0 = a0 ^ 1, b1 = a1 ^ 2, b2 = a2 ^ 3, b3 = a3 ^ 4, b4 = a4 ^ 5, b5 = a5 ^ 6, b6 = a6 ^ 7, b7 = a7 ^ 8;
    const uint32_t c = seed | 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
#define OP16(ins) asm volatile(ins " %0, %0, %16\n" ins " %1, %1, %16\n" ins " %2, %2, %16\n" ins " %3, %3, %16\n" ins " %4, %4, %16\n" ins " %5, %5, %16\n" ins " %6, %6, %16\n" ins " %7, %7, %16\n" \
                               ins " %8, %8, %16\n" ins " %9, %9, %16\n" ins " %10, %10, %16\n" ins " %11, %11, %16\n" ins " %12, %12, %16\n" ins " %13, %13, %16\n" ins " %14, %14, %16\n" ins " %15, %15, %16\n" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c))
        if (KIND == 0) OP16("v_add_u32");
        if (KIND == 1) OP16("v_xor_b32");
        if (KIND == 2) OP16("v_mul_lo_u32");
        if (KIND == 3) OP16("v_pk_max_u16");
        if (KIND == 4) OP16("v_lshlrev_b32");
        if (KIND == 5) OP16("v_pk_add_u16");
        if (KIND == 6) OP16("v_max_u32");
#define OPS16(ins, tail) asm volatile(ins " %0, %0, %16 " tail "\n" ins " %1, %1, %16 " tail "\n" ins " %2, %2, %16 " tail "\n" ins " %3, %3, %16 " tail "\n" ins " %4, %4, %16 " tail "\n" ins " %5, %5, %16 " tail "\n" ins " %6, %6, %16 " tail "\n" ins " %7, %7, %16 " tail "\n" \
                               ins " %8, %8, %16 " tail "\n" ins " %9, %9, %16 " tail "\n" ins " %10, %10, %16 " tail "\n" ins " %11, %11, %16 " tail "\n" ins " %12, %12, %16 " tail "\n" ins " %13, %13, %16 " tail "\n" ins " %14, %14, %16 " tail "\n" ins " %15, %15, %16 " tail "\n" \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c))
        if (KIND == 7) OPS16("v_max_u16_sdwa", "dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1");      // bytewise max in place (k_vote pass A candidate)
        if (KIND == 8) OPS16("v_perm_b32", ", %16");
        if (KIND == 9) OPS16("v_bfi_b32", ", %16");
        if (KIND == 10) OPS16("v_and_or_b32", ", %16");
        if (KIND == 11) OPS16("v_max_u16_sdwa", "dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:BYTE_2");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    const uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    if (r == 0x12345678u) *sink = r;
}
template <int KIND> void run(const char *name, unsigned long long *cyc, uint32_t *sink, int n_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 7, 8}) {
        const int blocks = n_cu * 4 * wps;                         // one-wave workgroups: wps waves on every SIMD when the dispatcher spreads them evenly
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, sink, 7u);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, cyc, sink, 7u);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double mean = 0; for (int k = 0; k < 64; k++) mean += (double)h[k]; mean /= 64;
        const double instr = (double)ITER * 16;
        printf("%-14s waves/SIMD %d: %.2f cycle-counter ticks per instruction of a wave; kernel %.3f ms -> %.1f G wave-instructions/s on the chip = %.3f per SIMD per ns\n",
               name, wps, mean / instr, ms, instr * blocks / ms / 1e6, instr * blocks / ms / 1e6 / (n_cu * 4));
    }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs, clockRate %d kHz, wall clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate, pr.clockInstructionRate);
    unsigned long long *cyc; uint32_t *sink; hipMalloc(&cyc, 8 * 65536); hipMalloc(&sink, 4);
    run<0>("v_add_u32", cyc, sink, pr.multiProcessorCount);
    run<1>("v_xor_b32", cyc, sink, pr.multiProcessorCount);
    run<2>("v_mul_lo_u32", cyc, sink, pr.multiProcessorCount);
    run<3>("v_pk_max_u16", cyc, sink, pr.multiProcessorCount);
    run<4>("v_lshlrev_b32", cyc, sink, pr.multiProcessorCount);
    run<5>("v_pk_add_u16", cyc, sink, pr.multiProcessorCount);
    run<6>("v_max_u32", cyc, sink, pr.multiProcessorCount);
    run<7>("v_max_u16_sdwa(byte)", cyc, sink, pr.multiProcessorCount);
    run<8>("v_perm_b32", cyc, sink, pr.multiProcessorCount);
    run<9>("v_bfi_b32", cyc, sink, pr.multiProcessorCount);
    run<10>("v_and_or_b32", cyc, sink, pr.multiProcessorCount);
    run<11>("v_max_u16_sdwa(w<-b)", cyc, sink, pr.multiProcessorCount);
    return 0;
}
