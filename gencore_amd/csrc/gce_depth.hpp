// gce_depth.hpp — Stats::statDepth + Bed::statDepth (src/stats.cpp:57-84, src/bed.cpp:66-81) over the stream that is resident in HBM
// (SURVEY.md 8(f)3): per-contig depth bins of `coverageStep` bases and per-BED-region base counts, once over every mapped input read
// (mPreStats->addRead, src/gencore.cpp:222 -> stats.cpp:118-120) and once over every emitted record (writeBam ->
// mPostStats->addRead, src/gencore.cpp:110).  A block per 2048 reads; the reference adds the read's l_qseq bases (not its reference
// span) from `pos` on, split over the bins it touches.  HBM-bound: 32 B key record in; the amounts meet in LDS first (below).
#pragma once
#include "gce_kernels.hpp"

struct DepthCtx {
    const int64_t *bin_off;            // [n_targets + 1]
    int32_t n_targets, step;
    const int32_t *reg_off;            // [n_targets + 1] regions of a contig (file order), CSR
    const int32_t *r_start, *r_end, *r_pmax;   // r_pmax: running maximum of r_end inside the contig
    const uint8_t *contig_sorted;      // [n_targets] region starts non-decreasing: the `break` of bed.cpp:75-76 only ends the scan
};

// Block-level aggregation in front of the global atomics (round 5): the stream is coordinate sorted, so the reads of a block fall into a handful of bins and
// regions -- on a capture panel thousands of consecutive reads add to the SAME bin and the same region, and a same-address atomic at the L2 is ~12 ns: the
// thread-per-read kernel with direct atomics took 6 ms for cfg3's 20 M reads.  Here every (bin | region, amount) goes into a small LDS hash table (64-bit LDS
// atomics); one global atomic per distinct key and block is left.  A key that finds no slot within a few probes goes to memory directly.
#define DP_T 256
#define DP_RPT 8             // reads per thread
#define DP_SLOTS 512
struct DepthAgg {
    unsigned long long *key, *val, *glob;
    __device__ __forceinline__ void add(uint64_t k, unsigned long long v) {
        uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 55) & (DP_SLOTS - 1);
#pragma unroll 1
        for (int probe = 0; probe < 8; probe++, h = (h + 1) & (DP_SLOTS - 1)) {
            const unsigned long long old = atomicCAS(&key[h], 0ull, (unsigned long long)k + 1ull);
            if (old == 0ull || old == k + 1ull) { atomicAdd(&val[h], v); return; }
        }
        atomicAdd(&glob[k], v);
    }
};

// sel == nullptr: every read of the batch with tid >= 0; else the reads sel[0..n)
__global__ __launch_bounds__(DP_T) void k_depth(const gce_core *core, const uint32_t *sel, uint64_t n, DepthCtx c, unsigned long long *depth, unsigned long long *bed) {
    __shared__ unsigned long long s_key[2][DP_SLOTS], s_val[2][DP_SLOTS];
    for (int i = threadIdx.x; i < 2 * DP_SLOTS; i += DP_T) { (&s_key[0][0])[i] = 0ull; (&s_val[0][0])[i] = 0ull; }
    __syncthreads();
    DepthAgg ad{s_key[0], s_val[0], depth}, ab{s_key[1], s_val[1], bed};
    const uint64_t base = (uint64_t)blockIdx.x * (DP_T * DP_RPT);
    for (int q = 0; q < DP_RPT; q++) {
        const uint64_t k = base + (uint64_t)q * DP_T + threadIdx.x;
        if (k >= n) break;
        const gce_core r = core[sel ? sel[k] : k];
        const int tid = r.tid, start = r.pos, len = r.l_qseq, end = start + len;
        if (tid < 0 || tid >= c.n_targets) continue;                                 // stats.cpp:118 (mapped only), :61-62
        // ---- BED regions (Stats::statDepth calls Bed::statDepth first, stats.cpp:58-59)
        const int rb = c.reg_off[tid], re = c.reg_off[tid + 1];
        if (re > rb) {
            if (c.contig_sorted[tid]) {
                int lo = rb, hi = re;                                                // first region with start > end
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.r_start[mid] > end) hi = mid; else lo = mid + 1; }
                for (int p = lo - 1; p >= rb && c.r_pmax[p] >= start; p--) {
                    const int pe = c.r_end[p];
                    if (pe < start) continue;                                        // bed.cpp:73-74
                    const int ps = c.r_start[p];
                    ab.add((uint64_t)p, (unsigned long long)(long long)(min(pe, end) - max(ps, start)));   // :78-79
                }
            } else {
                for (int p = rb; p < re; p++) {                                      // unsorted file: the literal loop, break and all
                    const int pe = c.r_end[p], ps = c.r_start[p];
                    if (pe < start) continue;
                    if (ps > end) break;
                    ab.add((uint64_t)p, (unsigned long long)(long long)(min(pe, end) - max(ps, start)));
                }
            }
        }
        // ---- genome bins (stats.cpp:64-83)
        const int64_t b0 = c.bin_off[tid], nb = c.bin_off[tid + 1] - b0;
        const int lp = start / c.step, rp = end / c.step;                            // C division: truncation toward zero, as the reference
        if (rp >= nb || lp < 0) continue;
        if (lp == rp) ad.add((uint64_t)(b0 + lp), (unsigned long long)(long long)len);
        else {
            ad.add((uint64_t)(b0 + lp), (unsigned long long)(long long)((lp + 1) * c.step - start));
            ad.add((uint64_t)(b0 + rp), (unsigned long long)(long long)(end - rp * c.step));
            for (int p = lp + 1; p < rp; p++) ad.add((uint64_t)(b0 + p), (unsigned long long)c.step);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DP_SLOTS; i += DP_T) {
        if (s_key[0][i]) atomicAdd(&depth[s_key[0][i] - 1ull], s_val[0][i]);
        if (s_key[1][i]) atomicAdd(&bed[s_key[1][i] - 1ull], s_val[1][i]);
    }
}
