// gce_deep.hpp — deep group sides (> 64 pairs: ultra-deep amplicons, BASELINE.json configs[4]): one BLOCK per (group, side).
//
// k_consensus_slow walks every voter per column on one wave (O(voters x columns) dependent byte loads); a side of 600 voters x 250
// columns keeps that wave busy for milliseconds while the chip idles.  Here the template pick and the voter list (side_prepare:
// CIGAR classes, group.cpp:136-315) still run on one wave, but the votes -- the O(voters x columns) part -- are spread over the
// block: every wave takes every 4th voter, a lane covers 4 adjacent columns of it with one 4-byte quality load, one 4-byte load of
// packed bases and (inside a mate-overlap patch) one 4-byte score load, DV_UNROLL voters in flight per wave.  The tallies of the
// five real bins (A, C, G, T, N) of every column live in LDS: one 64-bit atomic add of (count | score sum | quality sum) and one
// 32-bit atomic max (top quality) per vote.  Then a lane per column decides (decide_column: the same code as the <= 64 pair path),
// results are buffered in LDS, and the block writes the template back unless mismatchInc > 5 (group.cpp:537-558).
// Sides it cannot take -- a nibble outside A,C,G,T,N, a quality >= 128, a template longer than DV_COLS -- are left to
// k_consensus_slow untouched (gen_flag stays != 2).
#pragma once
#include "gce_kernels.hpp"

#define DV_T 256                 // threads per block
#define DV_COLS 512              // template columns per block
#define DV_CHUNK 256             // voters staged per round
#ifndef DV_UNROLL
#define DV_UNROLL 2          // rows of four voters in flight per wave
#endif
#define DV_DONE 2                // gen_flag value: side finished here

struct DVoter { uint64_t so, qo; int rl, ld; uint32_t patch, pad; };

// column -> LDS slot: a lane owns 16 adjacent columns, so one atomic instruction of a wave touches columns 16 apart: column j of every
// 16-column run goes to the j-th group of 32 slots (DV_COLS = 512 = 16 x 32), and the lanes land on consecutive slots
__device__ __forceinline__ int dv_slot(int col) { return ((col & 15) << 5) | (col >> 4); }

struct DeepRec { uint32_t e, out, nv, len_mode; };      // len | left_mode << 16

// template pick + voter list of every deep side, a wave per side (side_prepare keeps one wave busy; four sides per block keep the CU
// busy): sides k_vote_deep can take are appended to deep_list, sides without a template are finished here
__global__ __launch_bounds__(256) void k_deep_prepare(DevBatch b, DevParams p, Work w) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t n_slow = (uint32_t)w.si->n_slow;
    // sides differ by two orders of magnitude in size: a wave draws its next one when it is done (a fixed stride gave a few waves three big ones)
    if (blockIdx.x * WAVES_PER_BLOCK + wv >= n_slow) return;                         // (no more waves at the counter than sides: an empty list costs nothing)
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&w.si->prep_next, 1u);
        idx = (uint32_t)__shfl((int)idx, 0);
        if (idx >= n_slow) break;
        const uint32_t e = w.slow_list[idx], gi = e >> 1; const bool is_left = !(e & 1);
        const uint32_t begin = w.g_begin[gi], np = w.g_np[gi];
        if (np <= 64) continue;                                                   // here for another reason (exotic bases, long reads)
#ifdef DV_PROF
        const unsigned long long dp_t0_ = wall_clock64();
#endif
        const SidePrep sp = side_prepare(b, p, w, begin, np, is_left, lane);
#ifdef DV_PROF
        if (lane == 0) { atomicAdd(&w.si->prof[6], wall_clock64() - dp_t0_); atomicAdd(&w.si->prof[7], (unsigned long long)np * 100ull); atomicAdd(&w.si->prof[14], 1ull); }
#endif
        if (lane == 0) {
            if (sp.out == NONE32) { (is_left ? w.rp_left : w.rp_right)[gi] = NONE32; w.gen_flag[e] = DV_DONE; }
            else if (sp.len <= DV_COLS && sp.nv < 65536u) {
                DeepRec r; r.e = e; r.out = sp.out; r.nv = sp.nv; r.len_mode = (uint32_t)sp.len | (sp.left_mode ? 1u << 16 : 0u);
                ((DeepRec *)w.deep_list)[atomicAdd(&w.si->n_deep, 1u)] = r;
            }
        }
        WAVE_SYNC();
    }
}

#ifdef DV_PROF               // per-phase block time of k_vote_deep into the k_vote slots (build with -DVB_PROF -DDV_PROF): 0 setup, 1 staging, 2 votes, 3 flush, 4 decide, 5 write-back
#define DV_TICK(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&w.si->prof[k], now_ - dv_prev_); dv_prev_ = now_; } } while (0)
#else
#define DV_TICK(k) do { } while (0)
#endif
__global__ __launch_bounds__(DV_T) void k_vote_deep(DevBatch b, DevParams p, Work w) {
    __shared__ unsigned long long s_acc[5][DV_COLS];      // count (16) | biased score sum (24) << 16 | quality sum (24) << 40
    __shared__ uint32_t s_tq[5][DV_COLS];
    __shared__ DVoter s_v[DV_CHUNK];
    __shared__ uint8_t s_nb[DV_COLS], s_nq[DV_COLS];
    __shared__ int s_minc, s_exotic;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ uint32_t s_next;
    const uint32_t n_deep = w.si->n_deep;
    // a block draws its next side when it is done with one, from the END of the list: k_deep_prepare appends a side when it is prepared, the big ones last
    if (blockIdx.x >= n_deep) return;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_next = atomicAdd(&w.si->deep_next, 1u);
        __syncthreads();
        if (s_next >= n_deep) break;
        const uint32_t idx = n_deep - 1u - s_next;
        const DeepRec rec = ((const DeepRec *)w.deep_list)[idx];
        const uint32_t e = rec.e, gi = e >> 1; const bool is_left = !(e & 1);
        const uint32_t begin = w.g_begin[gi];
        __syncthreads();
#ifdef DV_PROF
        unsigned long long dv_prev_ = wall_clock64();
        if (threadIdx.x == 0) atomicAdd(&w.si->prof[15], 1ull);
#endif
        for (int k = tid; k < 5 * DV_COLS; k += DV_T) { (&s_acc[0][0])[k] = 0ull; (&s_tq[0][0])[k] = 0u; }
        if (tid == 0) { s_minc = 0; s_exotic = 0; }
        SidePrep sp; sp.out = rec.out; sp.nv = rec.nv; sp.len = (int)(rec.len_mode & 0xFFFFu); sp.left_mode = (rec.len_mode >> 16) & 1u;
        sp.voters = is_left ? w.pl : w.pu; sp.vld = is_left ? w.pr : w.members;      // side_prepare's scratch
        sp.ref = nullptr; sp.ref_len = 0;
        {
            const gce_core ok = b.core[sp.out];                                        // group.cpp:362-367 -> Reference::getData (reference.cpp:33-70)
            if (ok.isize != 0 && ok.tid >= 0 && ok.tid < p.n_ref) {
                const uint8_t *rd = p.ref_data[ok.tid];
                const int64_t need_len = (int64_t)d_ref_offset(b.cigar + b.cigar_off[sp.out], ok.n_cigar, sp.len - 1) + 1;
                if (rd && (int64_t)ok.pos + need_len < p.ref_len[ok.tid]) { sp.ref = rd; sp.ref_len = p.ref_len[ok.tid]; }
            }
        }
        uint32_t *rp_out = is_left ? w.rp_left : w.rp_right;
        const int len = sp.len;
        // ---- votes.  A lane takes SIXTEEN adjacent columns of a voter (two 8-byte quality loads, one 8-byte load of packed bases + the byte behind
        //      it for an odd start, two 8-byte score loads inside a mate-overlap patch), a wave four voters side by side (sub = lane / 16) and
        //      DV_UNROLL such rows in flight; the four waves interleave rows.  Votes FOR the template's base -- all but the sequencing errors --
        //      are tallied in registers, four columns per 32-bit word (counts in byte lanes; score / quality sums and top qualities in 16-bit
        //      lanes: a lane sees DV_CHUNK / 16 = 16 voters per chunk), and reach the LDS tallies once per chunk; only a vote for another base
        //      is an LDS atomic of its own.  Four columns per lane (round 2) spent ~125 VALU + ~100 SALU instructions per voter and wave on
        //      what is per-voter bookkeeping: descriptor, bounds, patch window, addresses.
        DV_TICK(0);
        const int sub = lane >> 4, ch = lane & 15;
        const uint8_t *tseq = b.seq + b.seq_off[rec.out];
        auto spread4 = [](uint32_t n16) { uint32_t x = n16 & 0xFFFFu; x = (x | (x << 8)) & 0x00FF00FFu; return (x | (x << 4)) & 0x0F0F0F0Fu; };   // nibble i -> byte i
        auto nib_order = [](uint64_t x) { return ((x & 0x0F0F0F0F0F0F0F0Full) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0Full); };                        // column i of the word at bits 4 i
        for (uint32_t qb = 0; qb < sp.nv; qb += DV_CHUNK) {
            __syncthreads();
            if (qb + tid < sp.nv) {
                const uint32_t r = sp.voters[begin + qb + tid];
                DVoter v; v.so = b.seq_off[r]; v.qo = b.qual_off[r]; v.rl = b.core[r].l_qseq; v.ld = (int)sp.vld[begin + qb + tid]; v.patch = w.spatch[r]; v.pad = 0;
                s_v[tid] = v;
            }
            __syncthreads();
            DV_TICK(1);
            const int lim = (int)min((uint32_t)DV_CHUNK, sp.nv - qb);
            for (int cb = 0; cb < len; cb += 256) {
                const int c0 = cb + 16 * ch;
                uint32_t tb[4] = {0, 0, 0, 0};                                          // the template's bases, one per byte
                if (c0 < len) {
                    const uint64_t y = nib_order(*(const u64_unaligned *)(tseq + (c0 >> 1)));
#pragma unroll
                    for (int g = 0; g < 4; g++) tb[g] = spread4((uint32_t)(y >> (16 * g)));
                }
                uint32_t C4[4] = {0, 0, 0, 0}, S02[4] = {0, 0, 0, 0}, S13[4] = {0, 0, 0, 0}, Q02[4] = {0, 0, 0, 0}, Q13[4] = {0, 0, 0, 0}, T02[4] = {0, 0, 0, 0}, T13[4] = {0, 0, 0, 0};
                for (int qr = 4 * wv; qr < lim; qr += 4 * (DV_T / 64) * DV_UNROLL) {     // (wave-uniform trips)
                    uint64_t qa[DV_UNROLL], qc[DV_UNROLL], sq[DV_UNROLL], sa[DV_UNROLL], sc[DV_UNROLL]; uint32_t s9[DV_UNROLL];
                    int rp0[DV_UNROLL]; bool on[DV_UNROLL], bytewise[DV_UNROLL], patched[DV_UNROLL];
#pragma unroll
                    for (int u = 0; u < DV_UNROLL; u++) {
                        const int q = qr + 4 * (DV_T / 64) * u + sub;
                        on[u] = q < lim && c0 < len; bytewise[u] = false; patched[u] = false; qa[u] = qc[u] = sq[u] = sa[u] = sc[u] = 0; s9[u] = 0; rp0[u] = 0;
                        if (on[u]) {
                            const DVoter v = s_v[q];
                            rp0[u] = sp.left_mode ? c0 : c0 + v.ld;
                            if (rp0[u] + 16 <= 0 || rp0[u] >= v.rl) on[u] = false;       // no column of this lane meets the read
                            else if (rp0[u] < 0) bytewise[u] = true;                     // (the first lane of a shorter right-aligned voter)
                            else {
                                const uint8_t *qp = b.qual + v.qo + rp0[u], *sp_ = b.seq + v.so + (rp0[u] >> 1);
                                qa[u] = *(const u64_unaligned *)qp; qc[u] = *(const u64_unaligned *)(qp + 8);
                                sq[u] = *(const u64_unaligned *)sp_; s9[u] = sp_[8];
                                const uint32_t pt = v.patch;
                                if (pt != GCE_PATCH_CONST && pt != 0u && rp0[u] < (int)((pt & 0xFFFF) + (pt >> 16)) && rp0[u] + 16 > (int)(pt & 0xFFFF)) {
                                    const uint8_t *cp = (const uint8_t *)w.score + v.qo + rp0[u];
                                    sa[u] = *(const u64_unaligned *)cp; sc[u] = *(const u64_unaligned *)(cp + 8); patched[u] = true;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < DV_UNROLL; u++) {
                        if (!on[u]) continue;
                        const int q = qr + 4 * (DV_T / 64) * u + sub;
                        const DVoter v = s_v[q];
                        if (bytewise[u]) {
                            for (int j = 0; j < 16; j++) {
                                const int col = c0 + j, rp = rp0[u] + j;
                                if (col >= len || rp < 0 || rp >= v.rl) continue;        // outside the voter: UB in the reference, skipped (as the oracle)
                                const int nib = d_nib(b.seq + v.so, rp), qu = b.qual[v.qo + rp];
                                const int scv = d_score_at(p, w.score + v.qo, v.patch, rp, qu);
                                const int k = nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : nib == 15 ? 4 : -1;
                                if (k < 0 || qu >= 128) { s_exotic = 1; continue; }
                                const int sl = dv_slot(col);
                                atomicAdd(&s_acc[k][sl], 1ull | ((unsigned long long)(unsigned)(scv + p.score_bias) << 16) | ((unsigned long long)(unsigned)qu << 40));
                                atomicMax(&s_tq[k][sl], (uint32_t)qu);
                            }
                            continue;
                        }
                        const int nval = min(min(16, len - c0), v.rl - rp0[u]);              // valid columns are j < nval (rp0 >= 0 here)
                        uint64_t y = nib_order(sq[u]);
                        if (rp0[u] & 1) y = (y >> 4) | ((uint64_t)(s9[u] >> 4) << 60);
                        const bool cst = v.patch == GCE_PATCH_CONST;
                        const int ws = (int)(v.patch & 0xFFFF), wl = (int)(v.patch >> 16);
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const int nv4 = nval - 4 * g;
                            if (nv4 <= 0) break;
                            const uint32_t vm = nv4 >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv4)) - 1u;
                            const uint32_t q4 = (uint32_t)((g < 2 ? qa[u] : qc[u]) >> (32 * (g & 1)));
                            if (q4 & 0x80808080u & vm) { s_exotic = 1; continue; }
                            const uint32_t vb4 = spread4((uint32_t)(y >> (16 * g)));
                            uint32_t sb4;                                                  // biased scores
                            if (cst) sb4 = 0x01010101u * (uint32_t)(p.s_moderate + p.score_bias);
                            else {
                                sb4 = d_q2s4_biased(p, q4 & 0x7F7F7F7Fu);
                                const int r4 = rp0[u] + 4 * g;
                                if (patched[u] && r4 < ws + wl && r4 + 4 > ws) {
                                    const uint32_t sc4 = (uint32_t)((g < 2 ? sa[u] : sc[u]) >> (32 * (g & 1)));
                                    uint32_t m = 0;
#pragma unroll
                                    for (int j = 0; j < 4; j++) if ((unsigned)(r4 + j - ws) < (unsigned)wl) m |= 0xFFu << (8 * j);
                                    sb4 = (sc4 & m) | (sb4 & ~m);
                                }
                            }
                            // bytes whose base is the template's (bases are < 16: + 15 reaches bit 4 iff they differ)
                            const uint32_t ne = (((vb4 ^ tb[g]) + 0x0F0F0F0Fu) >> 4) & 0x01010101u, eq1 = (ne ^ 0x01010101u) & vm;
                            const uint32_t m4 = (eq1 << 8) - eq1;
                            const uint32_t qm = q4 & m4, sm = sb4 & m4, qlo = qm & 0x00FF00FFu, qhi = (qm >> 8) & 0x00FF00FFu;
                            C4[g] += eq1;
                            S02[g] += sm & 0x00FF00FFu; S13[g] += (sm >> 8) & 0x00FF00FFu;
                            Q02[g] += qlo; Q13[g] += qhi;
                            T02[g] = pk_max_u16(T02[g], qlo); T13[g] = pk_max_u16(T13[g], qhi);
                            uint32_t rest = vm & ~m4 & 0x01010101u;                       // votes for another base: bit 8 j
                            while (rest) {
                                const int j = (__ffs((int)rest) - 1) >> 3;
                                rest &= rest - 1;
                                const uint32_t nib = (vb4 >> (8 * j)) & 15u;
                                const uint32_t k = (uint32_t)(0x4777777377727107ull >> (nib * 4)) & 7u;       // A,C,G,T,N -> 0..4, anything else 7
                                if (k == 7u) { s_exotic = 1; continue; }
                                const uint32_t qu = (q4 >> (8 * j)) & 0xFFu, sb = (sb4 >> (8 * j)) & 0xFFu;
                                const int sl = dv_slot(c0 + 4 * g + j);
                                atomicAdd(&s_acc[k][sl], (unsigned long long)(1u | (sb << 16)) | ((unsigned long long)(qu << 8) << 32));
                                atomicMax(&s_tq[k][sl], qu);
                            }
                        }
                    }
                }
                DV_TICK(2);
                // the lane's register tallies -> the template base's bins (a template base outside A,C,G,T,N: a voter with it is exotic)
                if (c0 < len) {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (C4[g] == 0u) continue;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t cnt = (C4[g] >> (8 * j)) & 0xFFu;
                            if (cnt == 0u) continue;
                            const uint32_t nib = (tb[g] >> (8 * j)) & 15u;
                            const uint32_t k = (uint32_t)(0x4777777377727107ull >> (nib * 4)) & 7u;
                            if (k == 7u) { s_exotic = 1; continue; }
                            const uint32_t ss = ((j & 1) ? S13[g] : S02[g]) >> (16 * (j >> 1)) & 0xFFFFu, qs = ((j & 1) ? Q13[g] : Q02[g]) >> (16 * (j >> 1)) & 0xFFFFu;
                            const uint32_t tq = ((j & 1) ? T13[g] : T02[g]) >> (16 * (j >> 1)) & 0xFFFFu;
                            const int sl = dv_slot(c0 + 4 * g + j);
                            atomicAdd(&s_acc[k][sl], (unsigned long long)(cnt | (ss << 16)) | ((unsigned long long)(qs << 8) << 32));
                            atomicMax(&s_tq[k][sl], tq);
                        }
                    }
                }
                DV_TICK(3);
            }
        }
        __syncthreads();
        if (s_exotic) continue;                                                    // nothing written yet: k_consensus_slow redoes the side
        // ---- a lane per column decides
        const uint32_t out = sp.out;
        const gce_core ok = b.core[out];
        const uint32_t *ocig = b.cigar + b.cigar_off[out];
        uint8_t *oseq = b.seq + b.seq_off[out], *oqual = b.qual + b.qual_off[out];
        int minc = 0;
        for (int col = tid; col < len; col += DV_T) {
            const int sl = dv_slot(col);
            Tally5 t; t.total = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const unsigned long long a = s_acc[k][sl];
                t.cnt[k] = (int)(a & 0xFFFFull); t.ss[k] = (int)((a >> 16) & 0xFFFFFFull) - t.cnt[k] * p.score_bias; t.qs[k] = (int)(a >> 40);
                t.tq[k] = (int)s_tq[k][sl];
                t.total += t.ss[k];
            }
            int ref4 = 0;
            if (sp.ref) {
                const int ro = d_ref_offset(ocig, ok.n_cigar, col);
                if (ro >= 0 && (int64_t)ok.pos + ro < sp.ref_len) ref4 = d_ref_nib(sp.ref, (int64_t)ok.pos + ro);
            }
            const int ob = d_nib(oseq, col);
            const ColOut r = decide_column(t, p, ob, ref4);
            s_nb[col] = (uint8_t)r.base; s_nq[col] = (uint8_t)r.qual;
            minc += r.minc;
        }
        if (minc) atomicAdd(&s_minc, minc);
        __syncthreads();
        DV_TICK(4);
        minc = s_minc;
        bool restore = false;
        if (minc != 0) {                                                           // group.cpp:528-573
            if (b.nm_type[out] == 0) { if (tid == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
            else if (minc > 5) restore = true;
            else if (tid == 0) { const int nn = b.nm[out] + minc; if (b.nm_type[out] == 'C' && nn >= 0 && nn <= 255) w.rp_nm[e] = nn; }
        }
        if (!restore) {
            for (int bi = tid; bi < (len + 1) / 2; bi += DV_T) {
                const int c0 = 2 * bi, c1 = c0 + 1;
                const uint8_t old = oseq[bi];
                oseq[bi] = (uint8_t)((s_nb[c0] << 4) | (c1 < len ? s_nb[c1] : (old & 0xF)));
            }
            for (int col = tid; col < len; col += DV_T) oqual[col] = s_nq[col];
        }
        if (tid == 0) { rp_out[gi] = out; w.gen_flag[e] = DV_DONE; }
        DV_TICK(5);
    }
}

// ===================================================================================================== deep clusters: pairing + UMI grouping
// One BLOCK per cluster of 65..PD_MAX reads (the generic path ranks its reads by an O(n^2 / 64) scan on a few waves and counts
// identical UMIs pair against pair: 15 ms of the 36 ms of configs[4]).  Everything lives in LDS:
//   1. reads ordered by (qname, input index) -- std::map<string, Pair*> order plus arrival order (cluster.cpp:260-273): a bitonic
//      sort of a u16 permutation; the comparator looks at 16 name bytes behind the cluster's common prefix (two big-endian words per
//      read), equal windows (mates, as a rule) fall back to strcmp and the read index
//   2. pairs: first read of a name run = left, last of the run (if any other) = right (pair.cpp:188-216), UMI agreement checked
//   3. UMI grouping (cluster.cpp:57-100): pairs sorted by their UMI words -> runs = distinct UMIs with their counts (umiCount);
//      the greedy loop (top count, lexicographically first on ties; absorb everything within the threshold) then runs over the
//      DISTINCT UMIs on one wave -- a few hundred entries, not thousands of pairs; umiDiff (cluster.cpp:41-53) is the number of
//      non-zero bytes of the XOR of the zero-padded words
//   4. layout: pairs sorted by (group, qname order) -> gpl / gpr, grp_begin / grp_n
// Clusters it cannot take (UMIs > 16 bytes, > PD_MAX reads) go to pq_list for the generic kernels.
#define PD_T 1024
#define PD_MAX 4096
#define PD_NONE16 0xFFFFu

// ascending bitonic sort of P (a power of two >= 128) entries.  Strides <= 64 stay inside one wave's 128-entry segment (thread t
// handles the pair (i, i + j) with i = 2 * (t & ~(j - 1)) | (t & (j - 1))), so only the strides >= 128 need a block barrier.
// strcmp of two strings of known lengths, 8 bytes at a time (big-endian words compare like the bytes)
__device__ __forceinline__ int pd_rest_cmp(const char *a, int la, const char *c, int lc) {
    for (int o = 0; o < la || o < lc; o += 8) {
        uint64_t x[1], y[1];
        load_be_words<1>(a + o, max(la - o, 0), x); load_be_words<1>(c + o, max(lc - o, 0), y);
        if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    }
    return 0;
}
template <bool GMEM, typename T, typename Less>
__device__ __forceinline__ void pd_bitonic(T *perm, int P, int tid, Less less) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += PD_T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                const bool up = (i & k) == 0;
                const T a = perm[i], c = perm[l];
                const bool swap = up ? less(c, a) : less(a, c);
                if (swap) { perm[i] = c; perm[l] = a; }
            }
            if (GMEM || j > 64 || j == 1 && (k << 1) > 128) __syncthreads(); else WAVE_SYNC();      // (arrays in memory: always the block barrier with its fences)
        }
}
// exclusive prefix over the block of one value per thread (+ total), s_w: 16 words of scratch
__device__ __forceinline__ uint32_t pd_scan(uint32_t v, uint32_t *s_w, int tid, uint32_t &total) {
    const int lane = tid & 63, wv = tid >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0; total = 0;
    for (int k = 0; k < PD_T / 64; k++) { const uint32_t x = s_w[k]; if (k < wv) base += x; total += x; }
    return base + inc - v;
}

#ifdef VB_PROF
#define PD_TICK(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&w.si->prof[16 + (k)], now_ - pd_prev_); pd_prev_ = now_; } } while (0)
#else
#define PD_TICK(k) do { } while (0)
#endif
// BIG = false: clusters of 65..PD_MAX reads, everything in LDS.  BIG = true: the clusters THAT one left behind for their size (up to 65 534
// reads: the 16-bit indices), the same algorithm with its arrays in a slab of device memory per block -- O(n log^2 n) instead of the generic
// kernels' O(n^2 / 64) for an ultra-deep hotspot; taken entries of pq_list are struck out (NONE32), the generic kernels skip them.
#define PD_BIGMAX 65536
#define PD_SLAB ((size_t)35 * PD_BIGMAX)       // bytes of device memory per block of the BIG instantiation
#define PD_BIG_BLOCKS 16
template <bool BIG>
__global__ __launch_bounds__(PD_T) void k_pairing_deep(DevBatch b, DevParams p, Work w, uint8_t *slab) {
    constexpr int MAXN = BIG ? PD_BIGMAX : PD_MAX, LN = BIG ? 2 : PD_MAX;
    __shared__ uint64_t l_key[LN][2];            // name windows of the reads, then UMI words of the pairs, then (u32) layout keys
    __shared__ uint16_t l_perm[LN];
    __shared__ uint16_t l_pl[LN], l_pr[LN];      // member index of the left / right read of pair i (qname order)
    __shared__ uint16_t l_pd[LN];                // pair -> distinct UMI, then pair -> group
    __shared__ uint32_t l_x[2 * LN + LN / 2];    // steps 1-2: read index and name pointer offset of every member; step 3: the distinct-UMI tables
    __shared__ uint8_t l_rest[LN];               // name bytes behind the window
    __shared__ uint32_t s_w[PD_T / 64];
    __shared__ int s_cp, s_flag, s_ngroups;
    uint64_t (*s_key)[2]; uint16_t *s_perm, *s_pl, *s_pr, *s_pd; uint32_t *s_x; uint8_t *s_rest;
    if (BIG) {
        uint8_t *m = slab + (size_t)blockIdx.x * PD_SLAB;
        s_key = reinterpret_cast<uint64_t (*)[2]>(m); m += (size_t)16 * MAXN;
        s_x = reinterpret_cast<uint32_t *>(m); m += (size_t)10 * MAXN;
        s_perm = reinterpret_cast<uint16_t *>(m); m += (size_t)2 * MAXN;
        s_pl = reinterpret_cast<uint16_t *>(m); m += (size_t)2 * MAXN;
        s_pr = reinterpret_cast<uint16_t *>(m); m += (size_t)2 * MAXN;
        s_pd = reinterpret_cast<uint16_t *>(m); m += (size_t)2 * MAXN;
        s_rest = m;
    } else { s_key = l_key; s_perm = l_perm; s_pl = l_pl; s_pr = l_pr; s_pd = l_pd; s_x = l_x; s_rest = l_rest; }
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ uint32_t s_li;
    const uint32_t n_list = BIG ? w.si->n_slow_pair2 : w.si->n_slow_pair;
    if (blockIdx.x >= n_list) return;
    for (;;) {                                                                     // (a block draws its next cluster when it is done with one: they differ tenfold in size)
        __syncthreads();
        if (tid == 0) { s_li = atomicAdd(BIG ? &w.si->pair_next2 : &w.si->pair_next, 1u); s_cp = 0x7FFFFFFF; s_flag = 0; }
        __syncthreads();
        const uint32_t li = s_li;
        if (li >= n_list) break;
        const uint32_t c = BIG ? w.left_list[li] : w.slow_list[li];
        const uint32_t start = w.cl_start[c], n = w.cl_n[c];
        bool take = BIG ? (n > PD_MAX && n <= PD_BIGMAX - 2) : (n > 64 && n <= PD_MAX);      // (BIG: only what the LDS instantiation left for its size)
        if (take) {
            int bad = 0;
            const uint64_t q0 = b.qname_off[w.members[start]];
            for (uint32_t i = tid; i < n; i += PD_T) {
                const uint32_t m = w.members[start + i];
                const int64_t dq = (int64_t)(b.qname_off[m] - q0);
                if (d_umi_len(w, m) > 16 || dq > 0x7FFF0000ll || dq < -0x7FFF0000ll) bad = 1;
            }
            if (bad) s_flag = 1;
        }
        __syncthreads();
        if (!take || s_flag) { if (!BIG && tid == 0) w.left_list[atomicAdd(&w.si->n_slow_pair2, 1u)] = c; continue; }      // (BIG: the entry stays for the generic kernels)
        const uint32_t mode = d_thr_mode(w.cl_ikey[c], w.si, p);
        if (mode == THR_NEVER) { if (tid == 0) { w.cl_npairs[c] = 0; w.cl_ngroups[c] = 0; w.cl_hasumi[c] = 0; if (BIG) w.left_list[li] = NONE32; } continue; }
#ifdef VB_PROF
        unsigned long long pd_prev_ = wall_clock64();
        if (threadIdx.x == 0) atomicAdd(&w.si->prof[30], 1ull);
#endif
        const int thr = mode == THR_PROPER ? p.proper_thr : p.unproper_thr;
        uint32_t *s_rd = s_x; uint32_t *s_qo = s_x + MAXN;
        uint16_t *s_dfirst = reinterpret_cast<uint16_t *>(s_x), *s_dcnt = s_dfirst + MAXN, *s_dgrp = s_dcnt + MAXN;
        // ---- 1. name windows + sort
        {
            const char *n0 = d_qname(b, w.members[start]);
            const int l0 = d_lqname(w, w.members[start]) - 1;
            int cp = 0x7FFFFFFF;
            for (uint32_t i = tid; i < n; i += PD_T) {
                const uint32_t m = w.members[start + i];
                const char *mq = d_qname(b, m);
                const int lim = min(l0, d_lqname(w, m) - 1);
                int l = 0;
                while (l + 8 <= lim) {                                               // 8 bytes at a time, then the first differing byte
                    const uint64_t x = *(const u64_unaligned *)(n0 + l) ^ *(const u64_unaligned *)(mq + l);
                    if (x) { l += (__ffsll((long long)x) - 1) >> 3; break; }
                    l += 8;
                }
                if (l + 8 > lim) while (l < lim && n0[l] == mq[l]) l++;
                cp = min(cp, l);
            }
            cp = wave_min(cp);
            if (lane == 0) atomicMin(&s_cp, cp);
        }
        __syncthreads();
        const int cp = s_cp;
        const char *qbase = b.qname + b.qname_off[w.members[start]];                 // name pointers of a cluster as 32-bit distances from here
        int P = 128; while (P < (int)n) P <<= 1;
        for (int i = tid; i < P; i += PD_T) {
            if (i < (int)n) {
                const uint32_t my = w.members[start + i];
                const int nl = d_lqname(w, my) - 1;
                uint64_t k2[2];
                load_be_words<2>(d_qname(b, my) + cp, max(nl - cp, 0), k2);
                s_key[i][0] = k2[0]; s_key[i][1] = k2[1];
                s_perm[i] = (uint16_t)i;
                s_rd[i] = my; s_rest[i] = (uint8_t)max(nl - cp - 16, 0); s_qo[i] = (uint32_t)(int32_t)((d_qname(b, my) + min(cp + 16, nl)) - qbase);      // where the window ends (checked above: fits)
            } else s_perm[i] = PD_NONE16;
        }
        __syncthreads();
        // The sort proper looks at LDS only: (window, read index).  Reads with equal windows (mates, as a rule) come out as one run in
        // arrival order; a run that holds different names behind the window -- rare -- is put right afterwards by one lane, with the
        // rest of the names compared 8 bytes at a time (inside the comparator those global loads were 60 % of the sort).
        auto name_less = [&](uint16_t a, uint16_t c2) -> bool {
            if (c2 == PD_NONE16) return a != PD_NONE16;
            if (a == PD_NONE16) return false;
            const uint64_t a0 = s_key[a][0], c0 = s_key[c2][0];
            if (a0 != c0) return a0 < c0;
            const uint64_t a1 = s_key[a][1], c1 = s_key[c2][1];
            if (a1 != c1) return a1 < c1;
            return s_rd[a] < s_rd[c2];
        };
        PD_TICK(0);
        pd_bitonic<BIG>(s_perm, P, tid, name_less);
        {
            int toolong = 0;
            for (int sidx = tid; sidx < (int)n; sidx += PD_T) {
                const uint16_t m = s_perm[sidx];
                const bool head = sidx == 0 || s_key[s_perm[sidx - 1]][0] != s_key[m][0] || s_key[s_perm[sidx - 1]][1] != s_key[m][1];
                if (!head || s_rest[m] == 0) continue;                              // (a name that ends inside the window has no rest: equal windows = equal names)
                int r = 1;
                while (sidx + r < (int)n && s_key[s_perm[sidx + r]][0] == s_key[m][0] && s_key[s_perm[sidx + r]][1] == s_key[m][1]) r++;
                if (r < 2) continue;
                if (r > 32) { toolong = 1; continue; }
                for (int x = 1; x < r; x++) {                                       // insertion sort of the run by (rest of the name, read index)
                    const uint16_t e = s_perm[sidx + x];
                    int y = x - 1;
                    for (; y >= 0; y--) {
                        const uint16_t o = s_perm[sidx + y];
                        const int cmp = pd_rest_cmp(qbase + (int32_t)s_qo[o], (int)s_rest[o], qbase + (int32_t)s_qo[e], (int)s_rest[e]);
                        if (cmp < 0 || (cmp == 0 && s_rd[o] < s_rd[e])) break;
                        s_perm[sidx + y + 1] = o;
                    }
                    s_perm[sidx + y + 1] = e;
                }
            }
            if (toolong) s_flag = 1;
        }
        __syncthreads();
        if (s_flag) { __syncthreads(); if (!BIG && tid == 0) w.left_list[atomicAdd(&w.si->n_slow_pair2, 1u)] = c; continue; }   // hundreds of names behind one window: generic kernels
        PD_TICK(1);
        // ---- 2. pairs
        const int per = (P + PD_T - 1) / PD_T;                                     // sorted positions per thread (contiguous)
        uint32_t firsts = 0;
        uint64_t isf_m = 0, isl_m = 0;                                             // (per <= 64: one bit per position of the thread)
        int any_umi = 0;
        for (int u = 0; u < per; u++) {
            const int sidx = tid * per + u;
            if (sidx < (int)n) {
                const uint16_t m = s_perm[sidx];
                const uint32_t q = s_rd[m];
                auto same = [&](uint16_t o) {
                    if (s_key[o][0] != s_key[m][0] || s_key[o][1] != s_key[m][1]) return false;
                    return pd_rest_cmp(qbase + (int32_t)s_qo[o], (int)s_rest[o], qbase + (int32_t)s_qo[m], (int)s_rest[m]) == 0;
                };
                const bool f_ = sidx == 0 || !same(s_perm[sidx - 1]), l_ = sidx == (int)n - 1 || !same(s_perm[sidx + 1]);
                isf_m |= (uint64_t)f_ << u; isl_m |= (uint64_t)l_ << u;
                if (!f_) {                                                     // setRight: UMI of the pair so far vs this read's (pair.cpp:201-212)
                    const uint32_t pv = s_rd[s_perm[sidx - 1]];
                    const int lp = d_umi_len(w, pv), lq = d_umi_len(w, q);
                    if (lp != 0) {
                        uint64_t x[2], y[2];
                        load_be_words<2>(d_umi_ptr(b, w, pv), lp, x); load_be_words<2>(d_umi_ptr(b, w, q), lq, y);     // (<= 16 bytes: checked above)
                        if (lp != lq || x[0] != y[0] || x[1] != y[1]) raise_error(w.si, GCE_ERR_UMI_MISMATCH, q);
                    }
                }
                if (l_ && d_umi_len(w, q)) any_umi = 1;
                firsts += f_;
            }
        }
        uint32_t npairs;
        uint32_t pbase = pd_scan(firsts, s_w, tid, npairs);
        if (any_umi) s_flag = 1;                                                   // (s_flag is 0 here)
        {
            uint32_t pi = pbase;
            // a run may straddle threads: the pair index of a non-first read is (firsts up to and including it) - 1
            for (int u = 0; u < per; u++) {
                const int sidx = tid * per + u;
                if (sidx < (int)n) {
                    const bool f_ = (isf_m >> u) & 1, l_ = (isl_m >> u) & 1;
                    if (f_) pi++;
                    const uint32_t pidx = pi - 1;
                    const uint16_t m = s_perm[sidx];
                    if (f_) { s_pl[pidx] = m; if (l_) s_pr[pidx] = PD_NONE16; }
                    if (l_ && !f_) s_pr[pidx] = m;
                }
            }
        }
        __syncthreads();
        any_umi = s_flag;
        __syncthreads();
        PD_TICK(2);
        // ---- 3. UMI grouping
        uint32_t ngroups = 1;
        if (!any_umi) { for (uint32_t i = tid; i < npairs; i += PD_T) s_pd[i] = 0; }
        else {
            int P2 = 128; while (P2 < (int)npairs) P2 <<= 1;
            for (int i = tid; i < P2; i += PD_T) {
                if (i < (int)npairs) {
                    const uint16_t m = s_pr[i] != PD_NONE16 ? s_pr[i] : s_pl[i];   // the pair's UMI is its last read's (pair.cpp:188-216)
                    const uint32_t ui = w.members[start + m];
                    uint64_t k2[2];
                    load_be_words<2>(d_umi_ptr(b, w, ui), (int)d_umi_len(w, ui), k2);
                    s_key[i][0] = k2[0]; s_key[i][1] = k2[1];
                    s_perm[i] = (uint16_t)i;
                } else s_perm[i] = PD_NONE16;
            }
            __syncthreads();
            auto umi_less = [&](uint16_t a, uint16_t c2) -> bool {
                if (c2 == PD_NONE16) return a != PD_NONE16;
                if (a == PD_NONE16) return false;
                const uint64_t a0 = s_key[a][0], c0 = s_key[c2][0];
                if (a0 != c0) return a0 < c0;
                const uint64_t a1 = s_key[a][1], c1 = s_key[c2][1];
                if (a1 != c1) return a1 < c1;
                return a < c2;
            };
            PD_TICK(3);
            pd_bitonic<BIG>(s_perm, P2, tid, umi_less);
            PD_TICK(4);
            const int per2 = (P2 + PD_T - 1) / PD_T;
            uint32_t heads = 0;
            isf_m = 0;
            for (int u = 0; u < per2; u++) {
                const int sidx = tid * per2 + u;
                if (sidx < (int)npairs) {
                    const uint16_t m = s_perm[sidx];
                    const bool head = sidx == 0 || s_key[s_perm[sidx - 1]][0] != s_key[m][0] || s_key[s_perm[sidx - 1]][1] != s_key[m][1];
                    isf_m |= (uint64_t)head << u; heads += head;
                }
            }
            uint32_t D;
            const uint32_t dbase = pd_scan(heads, s_w, tid, D);
            {
                uint32_t di = dbase;
                for (int u = 0; u < per2; u++) {
                    const int sidx = tid * per2 + u;
                    if (sidx < (int)npairs) {
                        if ((isf_m >> u) & 1) { di++; s_dfirst[di - 1] = s_perm[sidx]; s_dcnt[di - 1] = (uint16_t)sidx; s_dgrp[di - 1] = PD_NONE16; }   // dcnt: start for now
                        s_pd[s_perm[sidx]] = (uint16_t)(di - 1);
                    }
                }
            }
            __syncthreads();
            for (uint32_t d = tid; d < D; d += PD_T) {                             // run length = next start - own start
                const uint32_t nxt = d + 1 < D ? s_dcnt[d + 1] : npairs;
                s_dgrp[d] = (uint16_t)(nxt - s_dcnt[d]);                          // (parked in dgrp until every start was read)
            }
            __syncthreads();
            for (uint32_t d = tid; d < D; d += PD_T) { s_dcnt[d] = s_dgrp[d]; }
            __syncthreads();
            for (uint32_t d = tid; d < D; d += PD_T) s_dgrp[d] = PD_NONE16;
            __syncthreads();
            // the greedy loop over the DISTINCT UMIs.  Candidates in the order the reference would pick them if nothing were absorbed:
            // count descending, then UMI ascending (= distinct index ascending) -- one more sort, of (0xFFFF - count) << 16 | index
            uint32_t *ord = s_x + MAXN + MAXN / 2;                                           // behind the three u16 tables (3 x MAXN x 2 bytes); MAXN words
            int P4 = 128; while (P4 < (int)D) P4 <<= 1;
            for (int i = tid; i < P4; i += PD_T) ord[i] = i < (int)D ? ((0xFFFFu - (uint32_t)s_dcnt[i]) << 16 | (uint32_t)i) : 0xFFFFFFFFu;
            __syncthreads();
            pd_bitonic<BIG>(ord, P4, tid, [](uint32_t x, uint32_t y) { return x < y; });
            if (thr <= 0) {                                                        // nothing but the UMI itself is within 0: groups = candidates in order
                for (uint32_t r = tid; r < D; r += PD_T) s_dgrp[ord[r] & 0xFFFFu] = (uint16_t)r;
                if (tid == 0) s_ngroups = (int)D;
            } else {
                uint32_t ptr = 0, ng = 0;
                for (;;) {
                    while (ptr < D && s_dgrp[ord[ptr] & 0xFFFFu] != PD_NONE16) ptr++;      // (every thread walks the same way)
                    if (ptr >= D) break;
                    const uint32_t top = ord[ptr] & 0xFFFFu;
                    const uint64_t t0 = s_key[s_dfirst[top]][0], t1 = s_key[s_dfirst[top]][1];
                    __syncthreads();                                               // everybody has read the state of this round
                    for (uint32_t d = tid; d < D; d += PD_T) {
                        if (s_dgrp[d] != PD_NONE16) continue;
                        const uint64_t a0 = s_key[s_dfirst[d]][0], a1 = s_key[s_dfirst[d]][1];
                        if (popc_nonzero_bytes(a0 ^ t0) + popc_nonzero_bytes(a1 ^ t1) <= thr) s_dgrp[d] = (uint16_t)ng;
                    }
                    ng++;
                    __syncthreads();
                }
                if (tid == 0) s_ngroups = (int)ng;
            }
            __syncthreads();
            ngroups = (uint32_t)s_ngroups;
            for (uint32_t i = tid; i < npairs; i += PD_T) s_pd[i] = s_dgrp[s_pd[i]];
        }
        __syncthreads();
        PD_TICK(5);
        // ---- 4. layout: group by group, qname order kept inside a group (Group::addPair, group.cpp:17-22)
        {
            int P3 = 128; while (P3 < (int)npairs) P3 <<= 1;
            uint32_t *lk = reinterpret_cast<uint32_t *>(&s_key[0][0]);
            for (int i = tid; i < P3; i += PD_T) { lk[i] = i < (int)npairs ? ((uint32_t)s_pd[i] << 16 | (uint32_t)i) : 0xFFFFFFFFu; }
            __syncthreads();
            pd_bitonic<BIG>(lk, P3, tid, [](uint32_t a, uint32_t c2) { return a < c2; });
            PD_TICK(6);
            for (uint32_t sidx = tid; sidx < npairs; sidx += PD_T) {
                const uint32_t key = lk[sidx], g = key >> 16, i = key & 0xFFFFu;
                w.gpl[start + sidx] = w.members[start + s_pl[i]];
                w.gpr[start + sidx] = s_pr[i] != PD_NONE16 ? w.members[start + s_pr[i]] : NONE32;
                if (sidx == 0 || (lk[sidx - 1] >> 16) != g) { w.grp_begin[start + g] = start + sidx; s_pd[g] = (uint16_t)sidx; }   // (s_pd is free again: group -> first position)
            }
            __syncthreads();
            for (uint32_t sidx = tid; sidx < npairs; sidx += PD_T) {
                const uint32_t g = lk[sidx] >> 16;
                if (sidx + 1 == npairs || (lk[sidx + 1] >> 16) != g) w.grp_n[start + g] = sidx + 1 - (uint32_t)s_pd[g];
            }
        }
        PD_TICK(7);
        if (tid == 0) {
            const bool cross = d_key(b.core[w.members[start]], p).right < 0;
            w.cl_npairs[c] = npairs; w.cl_ngroups[c] = ngroups; w.cl_hasumi[c] = (uint8_t)((any_umi ? 1 : 0) | (cross ? 2 : 0));
            if (BIG) w.left_list[li] = NONE32;                                       // done here: struck out of the generic kernels' list
        }
    }
}
