// gce_depth.hpp — Stats::statDepth + Bed::statDepth (src/stats.cpp:57-84, src/bed.cpp:66-81) over the stream that is resident in HBM
// (SURVEY.md 8(f)3): per-contig depth bins of `coverageStep` bases and per-BED-region base counts, once over every mapped input read
// (mPreStats->addRead, src/gencore.cpp:222 -> stats.cpp:118-120) and once over every emitted record (writeBam ->
// mPostStats->addRead, src/gencore.cpp:110).  One thread per read; the reference adds the read's l_qseq bases (not its reference
// span) from `pos` on, split over the bins it touches.  HBM-bound: 32 B key record in, a few 64-bit atomics out; reads are
// coordinate sorted, so neighbouring lanes hit the same bins and L2 serves the atomics.
#pragma once
#include "gce_kernels.hpp"

struct DepthCtx {
    const int64_t *bin_off;            // [n_targets + 1]
    int32_t n_targets, step;
    const int32_t *reg_off;            // [n_targets + 1] regions of a contig (file order), CSR
    const int32_t *r_start, *r_end, *r_pmax;   // r_pmax: running maximum of r_end inside the contig
    const uint8_t *contig_sorted;      // [n_targets] region starts non-decreasing: the `break` of bed.cpp:75-76 only ends the scan
};

// sel == nullptr: every read of the batch with tid >= 0; else the reads sel[0..n)
__global__ __launch_bounds__(256) void k_depth(const gce_core *core, const uint32_t *sel, uint64_t n, DepthCtx c, unsigned long long *depth, unsigned long long *bed) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const gce_core r = core[sel ? sel[k] : k];
    const int tid = r.tid, start = r.pos, len = r.l_qseq, end = start + len;
    if (tid < 0 || tid >= c.n_targets) return;                                       // stats.cpp:118 (mapped only), :61-62
    // ---- BED regions (Stats::statDepth calls Bed::statDepth first, stats.cpp:58-59)
    const int rb = c.reg_off[tid], re = c.reg_off[tid + 1];
    if (re > rb) {
        if (c.contig_sorted[tid]) {
            int lo = rb, hi = re;                                                    // first region with start > end
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.r_start[mid] > end) hi = mid; else lo = mid + 1; }
            for (int p = lo - 1; p >= rb && c.r_pmax[p] >= start; p--) {
                const int pe = c.r_end[p];
                if (pe < start) continue;                                            // bed.cpp:73-74
                const int ps = c.r_start[p];
                atomicAdd(&bed[p], (unsigned long long)(long long)(min(pe, end) - max(ps, start)));   // :78-79
            }
        } else {
            for (int p = rb; p < re; p++) {                                          // unsorted file: the literal loop, break and all
                const int pe = c.r_end[p], ps = c.r_start[p];
                if (pe < start) continue;
                if (ps > end) break;
                atomicAdd(&bed[p], (unsigned long long)(long long)(min(pe, end) - max(ps, start)));
            }
        }
    }
    // ---- genome bins (stats.cpp:64-83)
    const int64_t nb = c.bin_off[tid + 1] - c.bin_off[tid];
    const int lp = start / c.step, rp = end / c.step;                                // C division: truncation toward zero, as the reference
    if (rp >= nb || lp < 0) return;
    unsigned long long *d = depth + c.bin_off[tid];
    if (lp == rp) atomicAdd(&d[lp], (unsigned long long)(long long)len);
    else {
        atomicAdd(&d[lp], (unsigned long long)(long long)((lp + 1) * c.step - start));
        atomicAdd(&d[rp], (unsigned long long)(long long)(end - rp * c.step));
        for (int p = lp + 1; p < rp; p++) atomicAdd(&d[p], (unsigned long long)c.step);
    }
}
