// gce_vote.hpp — Pair::computeScore + Group::consensusMergeBam + Group::makeConsensus for the usual groups, one WORKGROUP per batch of
// groups (pair.cpp:88-172, group.cpp:136-579).
//
// The per-group kernels of round 1 (one wave per group, k_score2 in front) were bound by their own instruction stream: ~1800 wave
// instructions per group, most of them bookkeeping executed with 6-19 of 64 lanes busy.  Here a workgroup of 256 threads takes a
// batch of ~9 groups (<= 16 groups, <= 95 pairs, weight 64) through a few FLAT phases, every phase with one lane per independent item:
//   P1  lane = pair            the two reads' descriptors into LDS; the pair's mate-overlap window (pair.cpp:108-120); what the sides'
//                              template choice needs of their reads (masks, position range) by LDS atomics
//   P2  lane = (pair, side), then lane = (group, side)   consensusMergeBam for the sides this kernel covers: one class of reads with the
//                              same CIGAR and length -- and position, when the right reads start at different positions (right-aligned
//                              mode) -- (+ a provably unrelated minority), template = first read of the class, voters = the class
//   P4  lane = (side, 16 columns)  "pass A": OR / AND of the voters' packed bases (unanimity), packed max of their quals; a column
//                              all voters agree on with top quality >= moderate takes group.cpp:421-428 (base kept, qual = max qual)
//   P3  lane = pair (behind pass A, in its phase: the bytes come from the L2)  mismatching bases in the mate overlap -> those columns are forced into the full vote of both sides
//   P5  lane = (side, voter, contested column)  "pass B": the voter's base, quality and exact score (pair.cpp:132-169 computed on the fly,
//                              the qualities of mismatching overlap bases rewritten to max(0, own - mate)) go into the column's 5-bin tally
//                              in LDS (count | biased score sum | quality sum packed into ONE atomic add, one atomic max for the top
//                              quality: 7.7 KB of tallies instead of 15 KB took the kernel from 5 to 7 waves per SIMD, -1.2 ms, after no
//                              change of the instruction stream had moved it); consecutive lanes = consecutive contested columns of ONE voter, so a wave's 64 byte loads fall
//                              into a dozen cache lines (column-major items made every lane touch a line of its own: the texture
//                              addresser, not HBM, was the limit).  Then lane = contested column: rule cascade + reference arbitration
//                              (group.cpp:394-501)
//   P6  lane = (group, side)   mismatchInc -> NM patch or restore (group.cpp:528-573), result records
//   P7  lane = (side, 16 columns)  write the template back
// Nothing is written to the reads before P7, and only the templates are: the quality rewrite of pair.cpp:158-159 exists in registers
// for the vote and persists exactly where the reference's persists in an emitted record (the template's columns are either voted
// columns, or — after a restore — rewritten here).  Scores are never materialised: no k_score2 launch, no score array traffic.
//
// A group with a side outside that scope (related odd reads, several classes of right reads on different positions, > 32 pairs, IUPAC codes, quals >= 128,
// unusual score constants ...) is handed on UNTOUCHED, both sides: gen_flag (-> k_consensus_fast / k_consensus_slow) and score_list
// (-> k_score2 scores just those pairs).
#pragma once

#ifdef VB_WAVE             // experiment: ONE wave per batch of weight 24 (8 groups, 16 sides): no workgroup barrier waits for another wave, the bytes pass A fetched are
#define VB_T 64            // voted on a few microseconds later
#define VB_W 24
#define VB_GDIV 8
#define VB_CCAP 96
#define VB_RCAP 160
#endif
#ifdef VB_BIG              // experiment: 512 threads per batch of weight 192 (32 groups, 64 sides: still one wave for the per-side phases), four workgroups per CU
#define VB_T 512
#define VB_W 192
#define VB_GDIV 32
#define VB_CCAP 512
#define VB_RCAP 768
#endif
#ifndef VB_GDIV
#define VB_GDIV 16
#endif
#ifndef VB_T
#define VB_T 256           // threads per batch
#endif
#ifndef VB_W
#define VB_W 96            // batch capacity in weight units (weight of a group = max(pairs, VB_MINW); a handed-on deep group takes a whole batch).  64 -> 96 in round 4:
                           // a batch's fixed phases and barriers carry half as many pairs again and pass A fills its lanes (k_vote 3.07 -> 2.89 ms)
#endif
static_assert(VB_W + 32 <= 256, "pair indices of a batch are bytes");
#define VB_MINW (VB_W / VB_GDIV) // smallest weight of a group: <= 16 (32) groups per batch (one lane per (group, side) in a wave)
#define VB_MAXG (VB_W / VB_MINW)
#define VB_MAXP (VB_W + 32)                // a batch's last group may reach over the end: < VB_W + 32 pairs
#define VB_SIDES (2 * VB_MAXG)
#define VB_COLS 256
#ifndef VB_CCAP
#define VB_CCAP 256        // (round 5: 160 -- 47 % of cfg3's batches held more than 128 contested columns and voted a second round, ~0.3 ms of the kernel; the LDS came from the voter lists, now 256 B, and the byte-wide s_wbase)  Round 4, at 128: (with VB_RCAP: LDS stays within 20 KB = EIGHT workgroups per CU, at 64 VGPRs (three dwords spilled): 2.79 -> 2.76 ms against
                           //  184 / 512 at seven (22.5 KB, 66 VGPRs), although nearly half of the batches now vote in two rounds; batches of weight 80 / 64 at eight
                           //  workgroups: 2.84 / 3.03 ms.  profiles/r04_q_ab_8waves.log)
#endif
//   VB_CCAP: contested columns voted per round (LDS tallies)
#ifndef VB_WPE
#define VB_WPE 8
#endif
#ifndef VB_TPLANE
#define VB_TPLANE 1             // tallies as ten planes of VB_CCAP dwords (add / max word of every bin), a pair item's columns half a side apart: consecutive lanes -> consecutive banks
#endif
#ifndef VB_COLPAIR
#define VB_COLPAIR 1          // pass-B items of TWO neighbouring contested columns of one voter (round 6); 0 = one column per item, two items per trip (rounds 2-5)
#endif
#if !VB_COLPAIR
#undef VB_TPLANE
#define VB_TPLANE 0
#endif
#ifndef VB_SMAX
#define VB_SMAX 32         // a side with more contested columns than this hands its group on
#endif
#ifndef VB_RCAP
#define VB_RCAP 384        // contested columns of a whole batch (all rounds): a side that does not fit any more hands its group on (32 sides x VB_SMAX would be 1024)
#endif

// A read of the batch in LDS, in two halves: the blob offsets live through the whole kernel; the rest (first CIGAR word, position, read index, length, CIGAR ops,
// fl bit 0: isize != 0) is what P1 .. P2 choose templates and voters by -- from pass A on its 4 KB are TALLY space (round 5: with them the tallies take 256
// contested columns in one round at the same 20 KB of LDS; at 128 nearly every batch of weight 96 voted two or three rounds, and a round costs every wave of
// the block its whole instruction stream however few items it has: 188 M of the kernel's 1067 M VALU instructions)
struct __attribute__((aligned(16))) VRead { uint64_t so, qo; };
struct __attribute__((aligned(16))) VTail { uint32_t c0; int32_t pos; uint32_t rd; uint16_t lq; uint8_t nc, fl; };
static_assert(sizeof(VRead) == 16 && sizeof(VTail) == 16, "VRead / VTail must stay 16 bytes");
__device__ __forceinline__ VTail vr_tail(const VTail *r) {                  // ONE LDS load
    union { uint4 q; VTail t; } u;
    u.q = *reinterpret_cast<const uint4 *>(r);
    return u.t;
}
struct VOv { uint16_t ls, rs, cmp, fl; };                   // fl: 1 = every score of the pair is the constant (pair.cpp:89-105), 2 = overlap [ls|rs, +cmp)
enum : uint8_t { VS_FINAL = 0, VS_ACTIVE = 1, VS_GEN = 2, VS_RESTORE = 3 };
struct __attribute__((aligned(8))) VSide {
    uint64_t ref; int64_t ref_len;
    uint32_t vmask, o_c0, result; int32_t o_pos, minc;
    uint16_t len, item0; uint8_t tmpl, nvot, o_nc, state, grp, lp0, pad[2];     // lp0: the group's first pair in the batch (s_rd index)
};

typedef unsigned short vb_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t c) {
    union { uint32_t u; vb_us2 v; } x, y, z; x.u = a; y.u = c; z.v = __builtin_elementwise_max(x.v, y.v); return z.u;
}
__device__ __forceinline__ uint64_t ld8_unaligned(const uint8_t *p_) { typedef uint64_t u64u __attribute__((aligned(1))); return *(const u64u *)p_; }
#ifndef VB_LD16
#define VB_LD16 1
#endif
typedef uint64_t vb_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vb_u64x2 vb_ld16(const uint8_t *p_) { typedef vb_u64x2 u128u __attribute__((aligned(1))); return *(const u128u *)p_; }   // one global_load_dwordx4
// gather the top bit of each of the 4 bytes of x into bits 0..3
__device__ __forceinline__ uint32_t msb4(uint32_t x) { return (((x >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu; }
// one bit per nibble of an 8-byte packed-base word (memory order: byte k = columns 2k (high nibble), 2k+1 (low nibble)) -> 16-bit column mask
__device__ __forceinline__ uint32_t nib_mask16(uint64_t nz /* bit 0 of every nibble */) {
    auto half = [](uint32_t v) { const uint32_t y = ((v >> 4) & 0x01010101u) | ((v << 1) & 0x02020202u); return ((y * 0x01041040u) >> 24) & 0xFFu; };
    return half((uint32_t)nz) | (half((uint32_t)(nz >> 32)) << 8);
}
__device__ __forceinline__ uint64_t nib_nonzero(uint64_t d) { return (d | (d >> 1) | (d >> 2) | (d >> 3)) & 0x1111111111111111ull; }
// nibbles that are one of the BAM codes 1, 2, 4, 8, 15 (A C G T N): popcount per nibble is 1 or 4
__device__ __forceinline__ uint64_t nib_acgtn(uint64_t v) {
    uint64_t t = v - ((v >> 1) & 0x5555555555555555ull);
    t = (t & 0x3333333333333333ull) + ((t >> 2) & 0x3333333333333333ull);        // popcount 0..4 in every nibble
    const uint64_t one = ~nib_nonzero(t ^ 0x1111111111111111ull), four = ~nib_nonzero(t ^ 0x4444444444444444ull);
    return (one | four) & 0x1111111111111111ull;
}

// d_ref_nib (gce_device.hpp) through a pointer that is known to be global memory
__device__ __forceinline__ int d_ref_nib_g(uint64_t ref, int64_t pos) {
    typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;
    const int bq = ((gptr_u8)ref)[pos >> 1];
    const int code = (pos & 1) ? (bq >> 4) : (bq & 0xF);
    return (int)((0x42810u >> (4 * (code < 5 ? code : 5))) & 0xFu);
}

// weights + batch starts: exclusive scan of the group weights (k_u64_reduce / k_u64_partials in front), first group of every batch
__global__ __launch_bounds__(256) void k_vote_batches(Work w, const unsigned long long *n_ptr, const uint64_t *part) {
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    const uint64_t n = *n_ptr, base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (base >= n) return;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = part[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        const uint64_t i = base + k * 256 + threadIdx.x;
        uint64_t v = i < n ? w.gw[i] : 0, x = v;
        for (int q = 1; q < 64; q <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, q); if (lane >= q) x += t; }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        const uint64_t carry = s_carry, ex = carry + woff + x - v;
        if (i < n) {
            w.g_wbase[i] = (uint32_t)ex;
            // the first group of a batch writes the batch's start: group i opens batch ex / VB_W iff no earlier group lies in it, i.e. iff the
            // group in front of it starts in an earlier batch (its start is ex - its weight, read from memory: NOT shuffled in from the
            // neighbour lane -- this branch is divergent at the end of the list, and a shuffle in a divergent branch reads garbage, which
            // is what a first version of this rule most likely did: a batch without start leaves its groups unvoted).  One store per batch
            // instead of one device-scope atomicMin per group: 1.56 M atomics were 57 of this kernel's 67 us.  Equivalent to the minimum
            // over the groups of the batch because the starts ascend with the index; groups of weight 0 do not exist (k_group_fill: >= 4).
            const bool first = i == 0 || (ex - w.gw[i - 1]) / VB_W != ex / VB_W;
            if (first) w.vb_start[ex / VB_W] = (uint32_t)i;
        }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + x;
        __syncthreads();
    }
}

__device__ __forceinline__ int vb_find(const uint16_t *pre, int n, int it) {       // largest s < n with pre[s] <= it  (pre ascending, pre[0] = 0)
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)pre[mid] <= it) lo = mid; else hi = mid - 1; }
    return lo;
}

// The same for the 64 consecutive items base + lane of a wave (all lanes must call it; items past `last` are clamped): the wave's first
// item is placed by one ballot over the <= 32 prefix entries, every lane then walks on from there FOUR entries per LDS round trip (the
// entries ascend: those <= the item are a prefix of the four) -- a wave's items span two to ten sides, and one entry per trip made
// the last lanes wait for as many dependent trips.
// No bounds tests (each one was an exec-mask branch of its own around one LDS read, four per trip): pre[n] is the total -- larger than every item, as
// is whatever follows it in the array (the next sides' prefixes, then the 0xFFFF the arrays are padded with: VB_PRE entries) -- so reading up to four
// entries past n is harmless and never counted.
#ifndef VB_FIND_ALIGNED
#define VB_FIND_ALIGNED 1
#endif
#define VB_PRE (VB_SIDES + 8)
__device__ __forceinline__ int vb_find_wave(const uint16_t *pre, int n, int base, int lane, int last) {
    last = max(last, 0);
    const int b0 = min(base, last), it = min(base + lane, last);
    const int pl = (int)pre[min(lane, n)];
    int s = __popcll(__ballot(pl <= b0)) - 1;
#if VB_FIND_ALIGNED
    // the four entries of the ALIGNED 8-byte group that holds pre[s + 1] (round 6: the walk read pre[s + 1 .. s + 4] -- an 8-byte LDS read on a 2-byte boundary three
    // times out of four; SQ_LDS_UNALIGNED_STALL was a third of the kernel's LDS-active cycles).  The group's entries up to s are <= the item like pre[s] itself, so
    // counting all four that are gives the place directly.
    for (int g = (s + 1) >> 2;; g++) {
        const uint2 w2 = *reinterpret_cast<const uint2 *>(pre + 4 * g);
        const int c = ((int)(w2.x & 0xFFFFu) <= it) + ((int)(w2.x >> 16) <= it) + ((int)(w2.y & 0xFFFFu) <= it) + ((int)(w2.y >> 16) <= it);
        if (c < 4) { s = 4 * g + c - 1; break; }
    }
#else
    for (;;) {
        const int a1 = (int)pre[s + 1], a2 = (int)pre[s + 2], a3 = (int)pre[s + 3], a4 = (int)pre[s + 4];
        const int c = (a1 <= it) + (a2 <= it) + (a3 <= it) + (a4 <= it);
        s += c;
        if (c < 4) break;
    }
#endif
    return s;
}

#if defined(VB_PROF) && !defined(DV_PROF)
#define VB_TICK(k) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&w.si->prof[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
#elif defined(VB_STOP)                  // cumulative cost of the phases (tools/vote_stop.sh): the kernel ends at tick VB_STOP
#define VB_TICK(k) do { if ((k) >= VB_STOP) return; } while (0)
#else
#define VB_TICK(k) do { } while (0)
#endif
#if defined(VB_PROF) && !defined(DV_PROF) && !defined(VB_COUNT)
#define VB_SUBTICK(k) VB_TICK(k)            // clocks inside a phase (slots 11.. of StreamInfo.prof; the phase's own slot then holds the rest)
#else
#define VB_SUBTICK(k) do { } while (0)
#endif
#define gb_ gb_
__global__ __launch_bounds__(VB_T) __attribute__((amdgpu_waves_per_eu(VB_WPE, 8))) void k_vote(DevBatch b, DevParams p, Work w, uint32_t n_groups) {
    __shared__ VRead s_rd[2][VB_MAXP];
    __shared__ VOv s_ov[VB_MAXP];
    __shared__ VSide s_side[VB_SIDES];
    __shared__ uint32_t s_cmask[VB_SIDES][VB_COLS / 32];
    __shared__ uint8_t s_vlist[2][VB_MAXP];                                        // voters of a side (pair index in the batch), ascending: the k-th voter of side (group, parity) at [parity][lp0 + k] -- a side has no more voters than its group has pairs
    __shared__ __attribute__((aligned(16))) uint32_t s_tal[VB_CCAP][5][2];         // pass B: per contested column and bin {count | biased score sum << 6 | qual sum << 20, top qual}:
                                                                                   // <= 32 voters, biased scores <= 255, quals < 128 on this path => 6 + 14 + 12 bits, one atomic add per vote
    __shared__ uint8_t s_ccol[VB_RCAP], s_cq[VB_RCAP], s_cb[VB_RCAP];              // contested columns (side by side, ascending): column; voted qual, voted base
    __shared__ __attribute__((aligned(8))) uint16_t s_jpre[VB_PRE];                                      // pass B: first (voter, column) item of every side
    __shared__ uint32_t s_ggi[VB_MAXG], s_gbeg[VB_MAXG];
    __shared__ __attribute__((aligned(8))) uint16_t s_ipre[VB_PRE], s_cpre[VB_PRE];                            // (entries behind VB_SIDES: 0xFFFF, see vb_find_wave)
    __shared__ uint8_t s_wbase[VB_SIDES][VB_COLS / 32];                            // place in the SIDE's contested-column list (<= VB_SMAX) of the first column of every 32-column word (P5b -> P7)
    __shared__ __attribute__((aligned(16))) uint16_t s_glp0[VB_MAXG];              // first pair of every group (unused entries: 0x7FFF)
    __shared__ uint8_t s_gnp[VB_MAXG], s_gflag[VB_MAXG];      // gflag: 1 = deep (handed on at once), 2 = odd / out of scope found later
    __shared__ int s_ng, s_np;
    __shared__ uint8_t s_ord[VB_SIDES];                                            // pass A: the sides in the order of their depth (P2b)
    __shared__ uint32_t s_cnt[VB_SIDES];                                           // contested columns of a side, counted by pass A (NOT in the tally space: the tallies are cleared while other waves still read these)
    // P1 -> P3 only, in the (not yet used) tally space: contig of either read of a pair (the template's reference lookup); length of its
    // last CIGAR op if that is an M block (isPartOf from the right end); per side the masks of its reads / its single-M reads, the range of
    // their positions, the voters and "a read outside the scope was seen"; the group of every pair
    int32_t (*s_ptid)[VB_MAXP] = reinterpret_cast<int32_t (*)[VB_MAXP]>(&s_tal[0][0][0]);
    uint16_t (*s_lastm)[VB_MAXP] = reinterpret_cast<uint16_t (*)[VB_MAXP]>(&s_tal[0][0][0] + 2 * VB_MAXP);
    uint32_t *s_hm = &s_tal[0][0][0] + 3 * VB_MAXP, *s_single = s_hm + VB_SIDES, *s_vm = s_single + VB_SIDES, *s_unf = s_vm + VB_SIDES;
    int32_t *s_pmin = reinterpret_cast<int32_t *>(s_unf + VB_SIDES), *s_pmax = s_pmin + VB_SIDES;
    uint8_t *s_pg = reinterpret_cast<uint8_t *>(s_pmax + VB_SIDES);
    // ... and, in its top 4 KB, the second half of every read (VTail) until P2 is through
    VTail (*s_rt)[VB_MAXP] = reinterpret_cast<VTail (*)[VB_MAXP]>(reinterpret_cast<char *>(&s_tal[0][0][0]) + sizeof(s_tal) - 2 * VB_MAXP * sizeof(VTail));
    static_assert(sizeof(s_tal) >= (3 * VB_MAXP + 6 * VB_SIDES) * 4 + VB_MAXP + 2 * VB_MAXP * sizeof(VTail), "P1 scratch and the reads' second halves must fit the tally space");
    // work index of a thread: rotated by the batch number, so that the single-wave phases (P0, P2, P5a, P6) and the half-empty ones
    // do not all land on the same SIMD of the CU (wave k of every workgroup runs on SIMD k)
#ifdef VB_XCD              // experiment: workgroup i runs on XCD i mod 8 -- give every XCD a CONTIGUOUS eighth of the batches, so that neighbouring batches (whose reads share sectors) share an L2
    const uint32_t per_x_ = gridDim.x >> 3, bid_ = (blockIdx.x & 7u) * per_x_ + (blockIdx.x >> 3);      // (the grid is a multiple of 8)
#else
    const uint32_t bid_ = blockIdx.x;
#endif
    const int tid = (int)((threadIdx.x + ((blockIdx.x & (VB_T / 64 - 1)) << 6)) & (VB_T - 1)), lane = tid & 63;
    const uint32_t g0 = w.vb_start[bid_];
    if (g0 == NONE32) return;
#if defined(VB_PROF) && !defined(DV_PROF)
    unsigned long long t_prev_ = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&w.si->prof[15], 1ull);
#endif
    // ---------------------------------------------------------------- P0: the groups of this batch
    if (tid < 64) {
        const uint32_t gi = g0 + (uint32_t)lane;
        const bool maybe = lane < VB_MAXG && gi < n_groups;                            // (the three loads side by side: whether the group belongs to the batch only decides who uses them)
        const uint32_t wb_ = maybe ? w.g_wbase[gi] : 0u, np_ = maybe ? w.g_np[gi] : 0u, gb_ = maybe ? w.g_begin[gi] : 0u;
        bool in = maybe && wb_ < (bid_ + 1u) * VB_W;
        uint32_t np = in ? np_ : 0u;
        const bool deep = in && (np > 32u || (int)np > p.skip_low_complexity_thr || !p.vote_ok);
        const unsigned long long im = __ballot(in);
        const int ng = __popcll(im);                                                   // (groups of a batch are consecutive: im = low bits)
        int x = deep ? 0 : (int)np, pre = x;
        pre = wave_scan_incl(pre);
        if (in) { s_ggi[lane] = gi; s_gbeg[lane] = gb_; s_gnp[lane] = (uint8_t)(deep ? 0u : np); s_glp0[lane] = (uint16_t)(pre - x); s_gflag[lane] = deep ? 1 : 0; }
        if (lane == ng - 1) s_np = pre;
        if (lane < VB_MAXG && !in) s_glp0[lane] = 0x7FFF;                              // (P1's packed search)
        if (lane == 0) s_ng = ng;
        if (deep) {                                                                    // both sides to the per-side kernels; k_score2 scores the group's pairs
            w.gen_flag[2 * gi] = 1; w.gen_flag[2 * gi + 1] = 1;
            const uint32_t gb = gb_;
            {   // one atomic per group: its pair slots go on k_score2's list, its two sides on gen_list
                const unsigned long long o_ = atomicAdd(&w.si->hand_on, (2ull << 32) | (unsigned long long)np);
                const uint32_t at = (uint32_t)o_, ga = (uint32_t)(o_ >> 32);
                w.gen_list[ga] = 2u * gi; w.gen_list[ga + 1] = 2u * gi + 1u;
                for (uint32_t k = 0; k < np; k++) w.score_list[at + k] = gb + k;
            }
            // the DEEP sides (consensus_fast_side's test) go on slow_list here and now: k_deep_prepare starts right behind this kernel, beside k_score2
            // (a pass of k_consensus_fast over gen_list just to find them was 0.38 ms of cfg5's critical path)
            if ((np > 64u || (int)np > p.skip_low_complexity_thr) && !(np == 1u && w.gpr[gb] == NONE32)) {
                const uint32_t at = atomicAdd(&w.si->n_slow, 2u);
                w.slow_list[at] = 2u * gi; w.slow_list[at + 1] = 2u * gi + 1u;
            }
        }
    }
    for (int k = tid; k < VB_SIDES * (VB_COLS / 32); k += VB_T) (&s_cmask[0][0])[k] = 0u;
    { const int t2 = VB_T > 64 ? tid - 64 : tid; if (t2 >= 0 && t2 < VB_PRE - VB_SIDES - 1) { const int k = VB_SIDES + 1 + t2; s_ipre[k] = 0xFFFF; s_cpre[k] = 0xFFFF; s_jpre[k] = 0xFFFF; } }      // (the second wave, where there is one)
    if (tid >= VB_T - 4 * VB_SIDES) {                                                  // (the last waves: the first one is busy with the groups)
        const int k = tid - (VB_T - 4 * VB_SIDES);
        s_hm[k] = 0u;                                                                   // s_hm, s_single, s_vm, s_unf are adjacent
        if (k < VB_SIDES) { s_pmin[k] = 0x7FFFFFFF; s_pmax[k] = -0x7FFFFFFF; }
    }
    __syncthreads();
    const int ng = s_ng, npairs = s_np;
    VB_TICK(0);
    // ---------------------------------------------------------------- P1: pairs -> read descriptors, overlap window
    if (tid < npairs) {
        // group of the pair = (number of groups that start at or in front of it) - 1: the starts (< 0x7FFF; unused entries hold 0x7FFF) as 16-bit
        // halves of LDS words, compared in pairs -- (0x8000 | tid) - start keeps bit 15 iff start <= tid, no borrow crosses a half
        int j = -1;
        {
            const uint32_t tb = 0x80008000u | (0x00010001u * (uint32_t)tid);
#pragma unroll
            for (int q = 0; q < VB_MAXG / 8; q++) {
                const uint4 g4 = reinterpret_cast<const uint4 *>(s_glp0)[q];
                j += __popc((tb - g4.x) & 0x80008000u) + __popc((tb - g4.y) & 0x80008000u) + __popc((tb - g4.z) & 0x80008000u) + __popc((tb - g4.w) & 0x80008000u);
            }
        }
        const uint32_t slot = s_gbeg[j] + (uint32_t)(tid - (int)s_glp0[j]);
        const uint32_t L = w.gpl[slot], R = w.gpr[slot];
        ReadDesc lk{}, rk{};
        if (L != NONE32) lk = load_desc(w.rdesc, L);
        if (R != NONE32) rk = load_desc(w.rdesc, R);
        VRead ol, orr; VTail vl, vr;
        ol.so = lk.so; ol.qo = lk.qo; vl.c0 = lk.c0; vl.pos = lk.pos; vl.rd = L; vl.lq = (uint16_t)lk.lq; vl.nc = (uint8_t)min((int)lk.nc, 255); vl.fl = lk.isize != 0;
        orr.so = rk.so; orr.qo = rk.qo; vr.c0 = rk.c0; vr.pos = rk.pos; vr.rd = R; vr.lq = (uint16_t)rk.lq; vr.nc = (uint8_t)min((int)rk.nc, 255); vr.fl = rk.isize != 0;
        s_rd[0][tid] = ol; s_rd[1][tid] = orr; s_rt[0][tid] = vl; s_rt[1][tid] = vr; s_ptid[0][tid] = lk.tid; s_ptid[1][tid] = rk.tid; s_lastm[0][tid] = lk.lastm; s_lastm[1][tid] = rk.lastm;
        s_pg[tid] = (uint8_t)j;
        {   // what the sides' template choice needs of their reads (group.cpp:177-194), gathered by the pairs
            const uint32_t kb = 1u << (tid - (int)s_glp0[j]);
            if (L != NONE32) { atomicOr(&s_hm[2 * j], kb); if (vl.nc == 1 && cig_op(vl.c0) == 0) atomicOr(&s_single[2 * j], kb); atomicMin(&s_pmin[2 * j], vl.pos); atomicMax(&s_pmax[2 * j], vl.pos); }
            if (R != NONE32) { atomicOr(&s_hm[2 * j + 1], kb); if (vr.nc == 1 && cig_op(vr.c0) == 0) atomicOr(&s_single[2 * j + 1], kb); atomicMin(&s_pmin[2 * j + 1], vr.pos); atomicMax(&s_pmax[2 * j + 1], vr.pos); }
        }
        VOv ov; ov.ls = 0; ov.rs = 0; ov.cmp = 0; ov.fl = 0;
        if (L == NONE32 || R == NONE32 || !(lk.ml > 0 && rk.ml > 0)) ov.fl = 1;        // pair.cpp:89-105: memset(scoreOfNotOverlappedModerateQual)
        else {
            const int dis = rk.pos - lk.pos;                                            // pair.cpp:108-120
            int ls, rs, cmp;
            if (dis >= 0) { ls = lk.mo + dis; rs = rk.mo; cmp = min(lk.ml - dis, rk.ml); }
            else { ls = lk.mo; rs = rk.mo - dis; cmp = min(lk.ml, rk.ml + dis); }
            if (cmp > 0) {
                if (lk.lq > 65535 || rk.lq > 65535) raise_error(w.si, GCE_ERR_INVALID, L);
                else { ov.ls = (uint16_t)ls; ov.rs = (uint16_t)rs; ov.cmp = (uint16_t)cmp; ov.fl = 2; }
            }
        }
        s_ov[tid] = ov;
    }
    __syncthreads();
    VB_TICK(1);
    // ---------------------------------------------------------------- P2: Group::consensusMergeBam per (group, side)   group.cpp:136-318
    // The side's majority class = CIGAR and length of its first single-M read.  The argument, on group.cpp:196-261: containedBy[i]
    // = 1 + #{j != i : isPartOf(read i, read j)}.  Identical reads (same CIGAR, same length) are part of each other, so every read of a
    // class C of n_c identical reads scores >= n_c.  A read that is UNRELATED to the class (isPartOf fails in both directions: it has >= 2
    // CIGAR ops and its first op is not an M block of >= len bases, while the class is one M block of len bases) scores at most the
    // number of unrelated reads, which is < n_c when the class is the strict majority.  The maximum is therefore taken by the class
    // reads, all with the same score and the same length, and the `>` / shorter-read scan of :235-258 keeps the FIRST of them: template
    // = first read of the class.  The voters of :287-313 are the reads isPartOf-compatible with the template = the class.  Right
    // side: only if all positions are equal (leftReadMode).  No single-M read at all: one class with the same 2-/3-op CIGAR and
    // nothing else.  Every side that is not of this shape hands its whole group on, untouched.
    // Right reads on different positions: not leftReadMode (group.cpp:177-194) -- isPartOf then compares CIGARs from their
    // END, containedBy only counts reads with the same right end (:220-224), columns align at the right end.  For one
    // class of identical reads (same CIGAR, length AND position) nothing changes, as long as every other read is unrelated
    // to it from the end as well: >= 2 ops and a LAST op that is not an M block of >= len bases (a leading soft clip,
    // typically) -- such a read neither votes (group.cpp:287-313) nor contains or is contained.
    // (a) lane = (pair, side): is my read of the template's class?  (one wave walking the <= 32 reads of its sides one after the other
    //     was a quarter of the batch's life time: 2 x 32 trips for the deepest group of the wave, at one wave's issue rate)
    for (int it = tid; it < 2 * npairs; it += VB_T) {
        const int pr = it >> 1, side = it & 1, jg = s_pg[pr], sl = 2 * jg + side, lp0 = s_glp0[jg];
        const VTail r = vr_tail(&s_rt[side][pr]);
        if (r.rd == NONE32 || s_gflag[jg] != 0) continue;
        const uint32_t hm = s_hm[sl], single = s_single[sl];
        const bool multi = single == 0;
        const int fl = multi ? __ffs((int)hm) - 1 : __ffs((int)single) - 1;
        const VTail t = vr_tail(&s_rt[side][lp0 + fl]);
        const bool ralign = side == 1 && s_pmin[sl] != s_pmax[sl];                     // (some read off the template's position <=> not all positions equal)
        uint32_t o_cw1 = 0, o_cw2 = 0, cw1 = 0, cw2 = 0;
        if (multi && t.nc >= 2 && t.nc <= 3) { const uint32_t *cg = b.cigar + b.cigar_off[t.rd]; o_cw1 = cg[1]; if (t.nc == 3) o_cw2 = cg[2]; }
        if (multi && r.nc >= 2 && r.nc <= 3) { const uint32_t *cg = b.cigar + b.cigar_off[r.rd]; cw1 = cg[1]; if (r.nc == 3) cw2 = cg[2]; }
        const int len = t.lq;
        const bool major = r.nc == t.nc && r.c0 == t.c0 && cw1 == o_cw1 && cw2 == o_cw2 && r.lq == t.lq && (!ralign || r.pos == t.pos);
        if (major) atomicOr(&s_vm[sl], 1u << (pr - lp0));
        else {
            // the op isPartOf looks at first: the first one, or the last when right aligned (its M length comes with the descriptor)
            const bool edge_m = (ralign && r.nc >= 2) ? (int)s_lastm[side][pr] >= len : (cig_op(r.c0) == 0 && cig_len(r.c0) >= len);
            if (multi || r.nc < 2 || edge_m) s_unf[sl] = 1u;
        }
    }
    __syncthreads();
    VB_SUBTICK(11);                                 // (-DVB_PROF: P2a alone)
    // (b) lane = (group, side)
    if (tid < 64) {
        const int j = lane >> 1, side = lane & 1;
        const bool mine = lane < 2 * ng && s_gflag[j] == 0;
        VSide sd; sd.ref = 0; sd.ref_len = 0; sd.vmask = 0; sd.o_c0 = 0; sd.result = NONE32; sd.o_pos = 0; sd.minc = 0; sd.len = 0; sd.item0 = 0;
        sd.tmpl = 0; sd.nvot = 0; sd.o_nc = 0; sd.state = VS_FINAL; sd.grp = (uint8_t)j; sd.lp0 = mine ? s_glp0[j] : 0; sd.pad[0] = sd.pad[1] = 0;
        bool to_gen = false;
        if (mine) {
            const int np = s_gnp[j], lp0 = s_glp0[j];
            const VTail *rds = s_rt[side] + lp0;
            // the reference of the side's contig (a side's reads share it, cross-contig clusters aside): asked for now, needed further down
            const int tid0 = s_ptid[side][lp0];
            const bool tid0_ok = tid0 >= 0 && tid0 < p.n_ref;
            const uint8_t *rdp0 = tid0_ok ? p.ref_data[tid0] : nullptr;
            const int64_t rl0 = tid0_ok ? p.ref_len[tid0] : 0;
            if (np == 1 && s_rt[1][lp0].rd == NONE32) {                                 // group.cpp:73-77: returned untouched
                sd.result = side == 0 ? s_rt[0][lp0].rd : NONE32;
            } else {
                const uint32_t hm = s_hm[lane], single = s_single[lane];
                if (hm != 0) {
                    const bool multi = single == 0;
                    const int fl = multi ? __ffs((int)hm) - 1 : __ffs((int)single) - 1;
                    const VTail t = vr_tail(rds + fl);
                    const int len = t.lq;
                    const bool ralign = side == 1 && s_pmin[lane] != s_pmax[lane];
                    const uint32_t vm = s_vm[lane];
                    const bool unfit = t.nc > 3 || t.nc < 1 || (ralign && multi) || s_unf[lane] != 0;
                    const int nvot = __popc(vm);
                    to_gen = unfit || nvot <= __popc(hm) - nvot || len > VB_COLS || len < 1;
                    if (!to_gen && !((double)nvot < (double)np * 0.4 && np != 1)) {    // group.cpp:264-266: else "no majority", result stays NONE
                        sd.state = VS_ACTIVE; sd.vmask = vm; sd.tmpl = (uint8_t)fl; sd.nvot = (uint8_t)nvot; sd.len = (uint16_t)len;
                        sd.o_pos = t.pos; sd.o_c0 = t.c0; sd.o_nc = t.nc; sd.result = t.rd;
                        if (t.fl & 1) {                                                 // group.cpp:362-367 -> Reference::getData (reference.cpp:33-70)
                            const int o_tid = s_ptid[side][lp0 + fl];
                            if (o_tid >= 0 && o_tid < p.n_ref) {
                                const uint8_t *rdp = o_tid == tid0 ? rdp0 : p.ref_data[o_tid];
                                const int64_t rl = o_tid == tid0 ? rl0 : p.ref_len[o_tid];
                                const int64_t need_len = (int64_t)(t.nc == 1 ? ((len - 1) < cig_len(t.c0) ? (len - 1) : -1) : d_ref_offset(b.cigar + b.cigar_off[t.rd], t.nc, len - 1)) + 1;
                                if (rdp && (int64_t)t.pos + need_len < rl) { sd.ref = (uint64_t)rdp; sd.ref_len = rl; }
                            }
                        }
                    }
                }
            }
        }
        VB_SUBTICK(12);                             // (-DVB_PROF: P2b up to the sides' states)
        if (lane < VB_SIDES) s_cnt[lane] = 0u;
        // a group goes on as a whole (all shuffles on wave-uniform paths)
        const int other_gen = __shfl_xor((int)to_gen, 1);                              // (unconditional: `a || shfl(..)` would shuffle in a divergent branch)
        const bool grp_gen = to_gen || other_gen != 0;
        if (mine && grp_gen) {
            sd.state = VS_GEN; sd.result = NONE32;
            if (side == 0) {
                const uint32_t gi = s_ggi[j];
                w.gen_flag[2 * gi] = 1; w.gen_flag[2 * gi + 1] = 1;
                s_gflag[j] = 2;                                                        // (P6 puts the group's pair slots on k_score2's list: once, with the groups found out of scope later)
            }
        }
        // pass-A items: 16-column chunks of the active sides, prefix over the sides
        // The sides in the order of their DEPTH (steps of four voters, deepest first): a pass-A wave runs to the deepest side among its 64 chunks, so sides of one
        // depth go to the same waves.  s_ord[k] = the side at place k; s_ipre runs over the places.  (Worth 2 % of the kernel's VALU instructions and 0.01 ms: a
        // batch's sides are closer to one another in depth than the clusters of the stream are.)
        const int nchunk = (mine && sd.state == VS_ACTIVE) ? (sd.len + 15) >> 4 : 0;
        const int cls = lane >= VB_SIDES ? -2 : nchunk ? min(7, ((int)sd.nvot + 3) / 4 - 1) : -1;
        const unsigned long long lt = (1ull << lane) - 1ull;
        int place = 0, cbase = 0;
#pragma unroll
        for (int c = 7; c >= -1; c--) {
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) place = cbase + __popcll(m & lt);
            cbase += __popcll(m);
        }
        if (lane < VB_SIDES) s_ord[place] = (uint8_t)lane;
        WAVE_SYNC();
        const int src = lane < VB_SIDES ? (int)s_ord[lane] : lane;                      // the side at my place
        const int nch_k = __shfl(nchunk, src);
        int pre = nch_k;
        pre = wave_scan_incl(pre);
        if (lane < VB_SIDES) { s_side[lane] = sd; s_ipre[lane] = (uint16_t)(pre - nch_k); }
        if (lane == VB_SIDES - 1) s_ipre[VB_SIDES] = (uint16_t)pre;
        WAVE_SYNC();
        if (lane < VB_SIDES) s_side[src].item0 = (uint16_t)(pre - nch_k);               // (behind the struct store of the side's own lane: LDS operations of one wave keep their order)
    }
    __syncthreads();
    VB_TICK(2);
    // the voter lists of a pair's two sides (ascending pair index): every voter knows its place from the side's mask  (read in P5)
    if (tid < npairs) {
        const int j = s_pg[tid], k = tid - (int)s_glp0[j];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const uint32_t vm = s_side[2 * j + side].vmask;                             // (0 unless the side is active)
            if ((vm >> k) & 1u) s_vlist[side][(int)s_glp0[j] + __popc(vm & ((1u << k) - 1u))] = (uint8_t)tid;     // (the pair's index in the batch)
        }
    }
    VB_TICK(3);
    // ---------------------------------------------------------------- P4: pass A, one lane per (side, 16 columns); the 16 top qualities stay
    //      in registers until the write-back (an item keeps its lane: it = tid + VB_T k, k < VB_IPL)
#define VB_IPL (VB_SIDES * (VB_COLS / 16) / VB_T)
    uint4 keep[VB_IPL];
    int item_side[VB_IPL];                                                         // side of every pass-A item of this lane (P4, P7)
    const int n_items = s_ipre[VB_SIDES];
#pragma unroll
    for (int kk = 0; kk < VB_IPL; kk++) {
        keep[kk] = make_uint4(0, 0, 0, 0);
        const int it = tid + VB_T * kk;
        item_side[kk] = 0;
        if (it - lane >= n_items) continue;                                        // (wave-uniform: the second round is empty for the usual batch of <= 256 items)
        const int s = s_ord[vb_find_wave(s_ipre, VB_SIDES, it - lane, lane, n_items - 1)];
        item_side[kk] = s;
        if (it < n_items) {
            const VSide sd = s_side[s];
            const int chunk = it - (int)sd.item0, c16 = 16 * chunk, nval = min(16, (int)sd.len - c16);
            const VRead *rds = s_rd[s & 1] + sd.lp0;
            uint64_t sor = 0, sand = ~0ull; uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
            // four voters per step: their twelve 8-byte loads are issued before the first byte is looked at; a short last step repeats the
            // step's first voter -- OR, AND and max do not mind
            for (uint32_t vm = sd.vmask; vm;) {
                const VRead *r0 = rds + (__ffs((int)vm) - 1); vm &= vm - 1;
                const VRead *r1 = vm ? rds + (__ffs((int)vm) - 1) : r0; vm &= vm ? vm - 1 : 0u;
                const VRead *r2 = vm ? rds + (__ffs((int)vm) - 1) : r0; vm &= vm ? vm - 1 : 0u;
                const VRead *r3 = vm ? rds + (__ffs((int)vm) - 1) : r0; vm &= vm ? vm - 1 : 0u;
                const uint64_t so0 = r0->so, qo0 = r0->qo, so1 = r1->so, qo1 = r1->qo, so2 = r2->so, qo2 = r2->qo, so3 = r3->so, qo3 = r3->qo;
                uint64_t sq[4], qa[4], qb[4];
#if VB_LD16        // the 16 qualities of a voter's chunk in ONE load (round 6: eight vector memory instructions per step instead of twelve)
                { const vb_u64x2 q0_ = vb_ld16(b.qual + qo0 + c16), q1_ = vb_ld16(b.qual + qo1 + c16), q2_ = vb_ld16(b.qual + qo2 + c16), q3_ = vb_ld16(b.qual + qo3 + c16);
                  sq[0] = ld8_unaligned(b.seq + so0 + 8 * chunk); sq[1] = ld8_unaligned(b.seq + so1 + 8 * chunk); sq[2] = ld8_unaligned(b.seq + so2 + 8 * chunk); sq[3] = ld8_unaligned(b.seq + so3 + 8 * chunk);
                  qa[0] = q0_.x; qb[0] = q0_.y; qa[1] = q1_.x; qb[1] = q1_.y; qa[2] = q2_.x; qb[2] = q2_.y; qa[3] = q3_.x; qb[3] = q3_.y; }
#else
                sq[0] = ld8_unaligned(b.seq + so0 + 8 * chunk); qa[0] = ld8_unaligned(b.qual + qo0 + c16); qb[0] = ld8_unaligned(b.qual + qo0 + c16 + 8);
                sq[1] = ld8_unaligned(b.seq + so1 + 8 * chunk); qa[1] = ld8_unaligned(b.qual + qo1 + c16); qb[1] = ld8_unaligned(b.qual + qo1 + c16 + 8);
                sq[2] = ld8_unaligned(b.seq + so2 + 8 * chunk); qa[2] = ld8_unaligned(b.qual + qo2 + c16); qb[2] = ld8_unaligned(b.qual + qo2 + c16 + 8);
                sq[3] = ld8_unaligned(b.seq + so3 + 8 * chunk); qa[3] = ld8_unaligned(b.qual + qo3 + c16); qb[3] = ld8_unaligned(b.qual + qo3 + c16 + 8);
#endif
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    sor |= sq[k]; sand &= sq[k];
                    const uint32_t q0 = (uint32_t)qa[k], q1 = (uint32_t)(qa[k] >> 32), q2 = (uint32_t)qb[k], q3 = (uint32_t)(qb[k] >> 32);
                    // bytewise maxima without unpacking: the HIGH byte of an unsigned 16-bit maximum is the maximum of the high bytes, whatever
                    // the low bytes hold -- the odd columns are the high bytes of q's halves as they come, the even ones after a shift by 8:
                    // three instructions per four columns (two v_perm + two v_pk_max_u16 before)
                    m0 = pk_max_u16(m0, q0 << 8); m1 = pk_max_u16(m1, q0);
                    m2 = pk_max_u16(m2, q1 << 8); m3 = pk_max_u16(m3, q1);
                    m4 = pk_max_u16(m4, q2 << 8); m5 = pk_max_u16(m5, q2);
                    m6 = pk_max_u16(m6, q3 << 8); m7 = pk_max_u16(m7, q3);
                }
            }
            // columns beyond the read (last chunk) read the neighbouring bytes: masked out of every test below
            const uint32_t colmask = nval >= 16 ? 0xFFFFu : ((1u << nval) - 1u);
            // (m even: columns 0, 2 of a word in bytes 1, 3; m odd: columns 1, 3 in bytes 1, 3)
            const uint32_t t0 = __builtin_amdgcn_perm(m1, m0, 0x07030501u), t1 = __builtin_amdgcn_perm(m3, m2, 0x07030501u);
            const uint32_t t2 = __builtin_amdgcn_perm(m5, m4, 0x07030501u), t3 = __builtin_amdgcn_perm(m7, m6, 0x07030501u);   // max quals, one byte per column
            const uint32_t differ = nib_mask16(nib_nonzero(sor ^ sand));
            const uint32_t valid = nib_mask16(nib_acgtn(sor));
            const uint32_t mod4 = 0x01010101u * (uint32_t)(p.moderate_q & 0x7F);
            auto ge4 = [&](uint32_t t) { return msb4(((t | 0x80808080u) - mod4)); };                 // per byte: t >= moderate (bytes < 128)
            const uint32_t geq = ge4(t0) | (ge4(t1) << 4) | (ge4(t2) << 8) | (ge4(t3) << 12);
            uint32_t contested = (differ | ~valid | ~geq) & colmask;
            if (!p.vote_accept_by_qual && (int)sd.nvot * p.s_min_lb < max(p.base_score_req, 1)) contested = colmask;   // scores cannot be bounded: vote everything
            // quals >= 128: out of scope (checked over the columns of the read only: the bytes behind a short last chunk are a neighbour's)
            const uint32_t hib = (msb4(t0) | (msb4(t1) << 4) | (msb4(t2) << 8) | (msb4(t3) << 12)) & colmask;
            if (hib != 0 || p.moderate_q > 127) s_gflag[sd.grp] = 2;
            keep[kk] = make_uint4(t0, t1, t2, t3);
            if (contested) {   // into the side's column mask, beside the columns P3 forces (in this phase, behind pass A); a bit is counted by whoever sets it first
                const uint32_t sh = 16u * (chunk & 1), mine_ = contested << sh, fresh = mine_ & ~atomicOr(&s_cmask[s][chunk >> 1], mine_);
                if (fresh) atomicAdd(&s_cnt[s], (uint32_t)__popc(fresh));
            }
        }
    }
    // ---------------------------------------------------------------- P3: mismatching bases in the mate overlap (pair.cpp:132-168)
    //      -> the column is forced into pass B on both sides (its scores are not qual2score(qual), its quals are rewritten).
    //      BEHIND pass A since round 5 (one lane per pair, the lanes that are through with their chunks): the two reads' bytes around the overlap were fetched by
    //      pass A a few microseconds ago and come from the L2 -- in front of P2, behind the descriptors, they were a third dependent trip to HBM in P1 and
    //      ~1 GB of sectors fetched twice (0.2 ms of the kernel in a build without the compare).  Pass A does not need the forced columns: a bit of s_cmask is
    //      counted by whoever sets it first (atomicOr returns the word as it was).
    //       This phase shares pass A's barrier interval: s_gflag[j] is read here while pass-A lanes of other waves may be setting a group's flag to 2, and both
    //      bump s_cnt behind first-setter atomicOrs on s_cmask.  The race is benign BY CONSTRUCTION and must stay so: a flag only ever goes 0 -> 2, a group flagged 2 is
    //      handed on untouched (P5a / P6 / P7 look at the flag again behind the barrier), so a stale 0 here only adds forced columns and counts to a side nobody votes on.
    //      Nothing behind this phase may use s_cnt or s_cmask of a flagged group.
    if (tid < npairs) {
        const VOv ov = s_ov[tid];
        const int j = s_pg[tid];
        if ((ov.fl & 2) && s_gflag[j] == 0) {
            const uint8_t *ls = b.seq + s_rd[0][tid].so, *rs = b.seq + s_rd[1][tid].so;
            const int lenl = s_side[2 * j].state == VS_ACTIVE ? (int)s_side[2 * j].len : 0, lenr = s_side[2 * j + 1].state == VS_ACTIVE ? (int)s_side[2 * j + 1].len : 0;
            // 8 columns of either read as nibbles in column order: swap the nibbles of every byte, drop the odd leading column
            auto cols8 = [](uint64_t x, int c0) {
                const uint64_t y = ((x & 0x0F0F0F0F0F0F0F0Full) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0Full);
                return (uint32_t)(y >> (4 * (c0 & 1)));
            };
            for (int i = 0; i < (int)ov.cmp; i += 32) {                                 // four 8-column words per step, loads first
                uint64_t lw[4], rw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ii = min(i + 8 * u, (int)ov.cmp - 1);                     // (clamped: a short last step re-reads a valid word)
                    lw[u] = ld8_unaligned(ls + ((ov.ls + ii) >> 1)); rw[u] = ld8_unaligned(rs + ((ov.rs + ii) >> 1));
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i0 = i + 8 * u, nv = min(8, (int)ov.cmp - i0);
                    if (nv <= 0) continue;
                    const int l0 = ov.ls + i0, r0 = ov.rs + i0;
                    uint32_t d = (cols8(lw[u], l0) ^ cols8(rw[u], r0)) & (nv >= 8 ? 0xFFFFFFFFu : ((1u << (4 * nv)) - 1u));
                    d = (d | (d >> 1) | (d >> 2) | (d >> 3)) & 0x11111111u;
                    while (d) {
                        const int k = (__ffs((int)d) - 1) >> 2;
                        d &= d - 1;
                        const int l = l0 + k, r = r0 + k;
                        if (l < VB_COLS) { const uint32_t bit = 1u << (l & 31); if (!(atomicOr(&s_cmask[2 * j][l >> 5], bit) & bit) && l < lenl) atomicAdd(&s_cnt[2 * j], 1u); }
                        if (r < VB_COLS) { const uint32_t bit = 1u << (r & 31); if (!(atomicOr(&s_cmask[2 * j + 1][r >> 5], bit) & bit) && r < lenr) atomicAdd(&s_cnt[2 * j + 1], 1u); }
                    }
                }
            }
        }
    }
    __syncthreads();
    VB_TICK(4);
    // ---------------------------------------------------------------- P5: pass B
    // (a) contested columns per side (counted by pass A); a side with too many hands its group on;
    //     prefixes over the sides: of the columns, and of the (voter, column) items
    if (tid < 64) {
        const bool act = lane < VB_SIDES && s_side[lane].state == VS_ACTIVE && s_gflag[s_side[lane].grp] == 0;
        int cnt = act ? (int)s_cnt[lane] : 0;
        const int over = cnt > VB_SMAX, other_over = __shfl_xor(over, 1);
        if (over && lane < VB_SIDES) s_gflag[s_side[lane].grp] = 2;
        if (over || other_over || !act) cnt = 0;
        int pre = cnt;
        pre = wave_scan_incl(pre);
        if (__any(pre > VB_RCAP)) {                                                     // (rare) the batch's column lists are full: the sides behind go on as well
            const int over2 = cnt > 0 && pre > VB_RCAP, other2 = __shfl_xor(over2, 1);
            if (over2 && lane < VB_SIDES) s_gflag[s_side[lane].grp] = 2;
            if (over2 || other2) cnt = 0;
            pre = cnt;
            pre = wave_scan_incl(pre);                                                  // (dropping sides only lowers the prefixes of the others)
        }
#if VB_COLPAIR
        const int nit = act ? ((cnt + 1) >> 1) * (int)s_side[lane].nvot : 0;          // (voter, column pair) items
#else
        const int nit = act ? cnt * (int)s_side[lane].nvot : 0;
#endif
        int pre2 = nit;
        pre2 = wave_scan_incl(pre2);
        if (lane < VB_SIDES) { s_cpre[lane] = (uint16_t)(pre - cnt); s_jpre[lane] = (uint16_t)(pre2 - nit); }
        if (lane == VB_SIDES - 1) { s_cpre[VB_SIDES] = (uint16_t)pre; s_jpre[VB_SIDES] = (uint16_t)pre2; }
    }
    __syncthreads();
    // (b) the list of contested columns, side by side: thread = (side, mask word)
    for (int q = tid; q < VB_SIDES * (VB_COLS / 32); q += VB_T) {
        const int s = q / (VB_COLS / 32), k = q % (VB_COLS / 32);
        if (s_side[s].state == VS_ACTIVE) {
            const int len = s_side[s].len;
            auto word = [&](int x) { uint32_t m = s_cmask[s][x]; const int hi = len - 32 * x; if (hi < 32) m &= hi <= 0 ? 0u : ((1u << hi) - 1u); return m; };   // forced columns behind the template's end do not exist
            const uint32_t mk = word(k);
            s_cmask[s][k] = mk;                                                         // (P7 reads the words again; a neighbour that still sees the unmasked word masks it itself)
            if (s_cpre[s + 1] > s_cpre[s]) {
                int base = s_cpre[s];
                for (int x = 0; x < k; x++) base += __popc(word(x));
                s_wbase[s][k] = (uint8_t)(base - (int)s_cpre[s]);
                for (uint32_t m = mk; m; m &= m - 1) s_ccol[base++] = (uint8_t)(32 * k + __ffs((int)m) - 1);
            }
        }
    }
    const int n_cont = s_cpre[VB_SIDES];
    // sides are voted in rounds of whole sides whose columns fit the tallies (usually one round)
    for (int s0 = 0; s0 < VB_SIDES;) {
        // sides [s0, s1): the longest run whose columns fit (a side has <= VB_SMAX of them).  One look per lane and a ballot -- the walk
        // `while (s_cpre[s1 + 1] - s_cpre[s0] <= VB_CCAP) s1++` was 32 dependent LDS round trips in every wave of the block
        const int fits_ = lane < VB_SIDES && (lane < s0 || (int)s_cpre[lane + 1] - (int)s_cpre[s0] <= VB_CCAP);
        const unsigned long long nofit_ = ~__ballot(fits_);
        const int s1 = nofit_ ? __ffsll((long long)nofit_) - 1 : 64;                            // first side that does not fit (prefixes ascend: every later one does not either); VB_SIDES if all do
        const int c0 = s_cpre[s0], ncol = (int)s_cpre[s1] - c0, j0 = s_jpre[s0], njob = (int)s_jpre[s1] - j0;
#if VB_TPLANE
        {
            uint32_t *tp_ = &s_tal[0][0][0];
            for (int c = tid; c < ncol; c += VB_T) {
#pragma unroll
                for (int pl = 0; pl < 10; pl++) tp_[pl * VB_CCAP + c] = 0u;
            }
        }
#else
        for (int k = tid; k < ncol * 5; k += VB_T) *(uint2 *)(&s_tal[0][0][0] + 2 * k) = make_uint2(0, 0);
#endif
        __syncthreads();
        VB_TICK(5);
#if VB_COLPAIR
        // (c) one lane per (side, voter, PAIR of neighbouring contested columns of the side's list), column pairs fastest (round 6).  What an item pays before it can ask for
        //     its bytes -- its side in the prefix, the division by the side's width, the voter, the voter's blob offsets and overlap window -- belongs to the VOTER, not to the
        //     column: two columns share it (rounds 2-5: one column per item, 144 wave instructions per 64 votes of which ~50 are the vote itself; voter quads x one column, round 5,
        //     shared only the side).  A side with an odd number of columns pads half an item per voter.  The byte loads of both columns are in flight together.
        //     k_vote executes 7 % fewer VALU instructions (1.029 -> 0.954 G per launch at cfg3, profiles/r06_e_*).
        struct Item { int ci, d1, side, grp; int q[2], sb[2], mb[2], mq[2], mc[2], col[2]; bool on, on1, cst; bool inov[2]; };      // (d1: the second column's distance in the list)
        auto prep = [&](int it, int s) -> Item {
            Item x; x.on = it < j0 + njob; x.on1 = false; x.ci = 0; x.d1 = 0; x.side = 0; x.grp = 0; x.cst = false;
#pragma unroll
            for (int u = 0; u < 2; u++) { x.q[u] = 0; x.sb[u] = 0; x.mb[u] = 0; x.mq[u] = 0; x.mc[u] = 0; x.col[u] = 0; x.inov[u] = false; }
            if (!x.on) return x;
            x.side = s & 1;
            const int cb = (int)s_cpre[s], ncs = (int)s_cpre[s + 1] - cb, ncs2 = (ncs + 1) >> 1, local = it - (int)s_jpre[s];
#if VB_TPLANE
            const int kv = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ncs2)), c = local - kv * ncs2;            // local / ncs2 (local < 512, ncs2 <= 16: the half keeps it exact)
            x.ci = cb + c; x.on1 = c + ncs2 < ncs; x.d1 = x.on1 ? ncs2 : 0;                                                  // the item's columns: c and c + half the side's width
            x.col[0] = s_ccol[x.ci]; x.col[1] = s_ccol[x.ci + x.d1];
#else
            const int kv = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ncs2)), c = 2 * (local - kv * ncs2);      // local / ncs2 (local < 512, ncs2 <= 16: the half keeps it exact)
            x.ci = cb + c; x.on1 = c + 1 < ncs; x.d1 = x.on1 ? 1 : 0;
            x.col[0] = s_ccol[x.ci]; x.col[1] = s_ccol[x.ci + x.d1];
#endif
            const VSide *sd = &s_side[s];
            x.grp = sd->grp;
            const int lp = s_vlist[x.side][(int)sd->lp0 + kv];
            const VRead *r = &s_rd[x.side][lp];
            const uint64_t so = r->so, qo = r->qo;
            const VOv ov = s_ov[lp];
            const int mystart = x.side ? ov.rs : ov.ls, matestart = x.side ? ov.ls : ov.rs;
            x.cst = ov.fl & 1;
            bool any_ov = false;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                x.inov[u] = (ov.fl & 2) && (unsigned)(x.col[u] - mystart) < (unsigned)ov.cmp;
                any_ov |= x.inov[u];
                x.sb[u] = b.seq[so + (x.col[u] >> 1)];
                x.q[u] = b.qual[qo + x.col[u]];
            }
            if (any_ov) {                                                               // pair.cpp:132-168: the mate's base and quality on the same reference position
                const VRead *mt = &s_rd[x.side ^ 1][lp];
                const uint64_t mso = mt->so, mqo = mt->qo;
#pragma unroll
                for (int u = 0; u < 2; u++) if (x.inov[u]) {
                    x.mc[u] = x.col[u] - mystart + matestart;
                    x.mb[u] = b.seq[mso + (x.mc[u] >> 1)]; x.mq[u] = b.qual[mqo + x.mc[u]];
                }
            }
            return x;
        };
        auto vote1 = [&](const Item &x, int u) {
            int q = x.q[u];
            const int nb = (x.col[u] & 1) ? (x.sb[u] & 0xF) : (x.sb[u] >> 4);
            const int mn = (x.mc[u] & 1) ? (x.mb[u] & 0xF) : (x.mb[u] >> 4);
            const bool match = x.inov[u] && nb == mn, mism = x.inov[u] && nb != mn;
            const bool left_wins = x.side ? (x.mq[u] >= q) : (q >= x.mq[u]);            // `if(lq >= rq)`: the left read keeps a score
            const int dq = max(0, q - x.mq[u]);
            const int qlook = match ? (((q + x.mq[u]) / 2) & 0xFF) : mism ? dq : q;
            const bool scored = !mism || (x.side == 0 ? left_wins : !left_wins);        // the mismatch loser scores 0
            // Pair::qual2score (pair.cpp:77-86) by the NUMBER OF THRESHOLDS PASSED (they are nested on this path: vote_ok) into the four biased scores packed in q2s_lut.
            // d_qual2score's nested selects compile to a select of ADDRESSES and one vector load from the kernel-argument segment: a third dependent trip to memory in every
            // item (bytes -> score -> tally), 2.4 us of a batch's 9.6 us in this phase (-DVB_PROF, profiles/r06_*).
            const int nthr = (qlook >= p.low_q) + (qlook >= p.moderate_q) + (qlook >= p.high_q);
            int sc = (int)((p.q2s_lut >> (8 * nthr)) & 0xFFu) + (match ? 4 : mism ? -3 : 0);          // biased
            sc = scored ? sc : p.score_bias;
            sc = x.cst ? p.s_moderate + p.score_bias : sc;
            q = mism ? dq : q;                                                          // the rewritten quality is what the vote sees
            const int bin = (int)((uint32_t)(0x4777777377727107ull >> (nb * 4)) & 7u);        // A,C,G,T,N -> 0..4, anything else 7
            if (bin == 7 || (q & 0x80)) s_gflag[x.grp] = 2;
            else {
#if VB_TPLANE
                uint32_t *t2 = &s_tal[0][0][0] + (2 * bin) * VB_CCAP + (x.ci + (u ? x.d1 : 0) - c0);
                atomicAdd(t2, 1u | ((uint32_t)sc << 6) | ((uint32_t)q << 20)); atomicMax(t2 + VB_CCAP, (uint32_t)q);
#else
                uint32_t *t2 = &s_tal[x.ci + (u ? x.d1 : 0) - c0][bin][0];
                atomicAdd(t2, 1u | ((uint32_t)sc << 6) | ((uint32_t)q << 20)); atomicMax(t2 + 1, (uint32_t)q);
#endif
            }
        };
        for (int itb = j0 + tid - lane; itb < j0 + njob; itb += VB_T) {                 // (wave-uniform trips: the side lookup is a wave operation)
            const int sa = s0 + vb_find_wave(s_jpre + s0, s1 - s0, itb, lane, j0 + njob - 1);
            const Item x0 = prep(itb + lane, sa);
            if (x0.on) { vote1(x0, 0); if (x0.on1) vote1(x0, 1); }
        }
#else
        // (c) one lane per (side, voter, contested column), columns fastest.  Two items per trip: the byte loads of both are issued
        //     before either is used (an item is two dependent round trips otherwise: LDS lookups -> its bytes)
        struct Item { int ci, side, grp, q, sb, mb, mq, mc, col; bool on, inov, cst; };
        auto prep = [&](int it, int s) -> Item {
            Item x; x.on = it < j0 + njob; x.ci = 0; x.side = 0; x.grp = 0; x.q = 0; x.sb = 0; x.mb = 0; x.mq = 0; x.mc = 0; x.col = 0; x.inov = false; x.cst = false;
            if (!x.on) return x;
            x.side = s & 1;
            const int ncs = (int)s_cpre[s + 1] - (int)s_cpre[s], local = it - (int)s_jpre[s];
            const int kv = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)ncs)), c = local - kv * ncs;      // local / ncs (local < 1024, ncs <= 32: the half keeps it exact)
            x.ci = (int)s_cpre[s] + c; x.col = s_ccol[x.ci];
            const VSide *sd = &s_side[s];
            x.grp = sd->grp;
            const int lp = s_vlist[x.side][(int)sd->lp0 + kv];
            const VRead *r = &s_rd[x.side][lp];
            const VOv ov = s_ov[lp];
            const int mystart = x.side ? ov.rs : ov.ls;
            x.cst = ov.fl & 1;
            x.inov = (ov.fl & 2) && (unsigned)(x.col - mystart) < (unsigned)ov.cmp;
            x.sb = b.seq[r->so + (x.col >> 1)];
            x.q = b.qual[r->qo + x.col];
            if (x.inov) {                                                               // pair.cpp:132-168: the mate's base and quality on the same reference position
                const VRead *mt = &s_rd[x.side ^ 1][lp];
                x.mc = x.col - mystart + (x.side ? ov.ls : ov.rs);
                x.mb = b.seq[mt->so + (x.mc >> 1)]; x.mq = b.qual[mt->qo + x.mc];
            }
            return x;
        };
        auto vote = [&](const Item &x) {
            if (!x.on) return;
            int q = x.q;
            const int nb = (x.col & 1) ? (x.sb & 0xF) : (x.sb >> 4);
            // the three score rules without branches (pair.cpp:89-105 constant; :140-150 match; :151-168 mismatch): one qual2score of the
            // quality each rule looks at, then the rule's offset
            const int mn = (x.mc & 1) ? (x.mb & 0xF) : (x.mb >> 4);
            const bool match = x.inov && nb == mn, mism = x.inov && nb != mn;
            const bool left_wins = x.side ? (x.mq >= q) : (q >= x.mq);                // `if(lq >= rq)`: the left read keeps a score
            const int dq = max(0, q - x.mq);
            const int qlook = match ? (((q + x.mq) / 2) & 0xFF) : mism ? dq : q;
            const bool scored = !mism || (x.side == 0 ? left_wins : !left_wins);        // the mismatch loser scores 0
            int sc = d_qual2score(p, qlook) + (match ? 4 : mism ? -3 : 0);
            sc = scored ? sc : 0;
            sc = x.cst ? p.s_moderate : sc;
            q = mism ? dq : q;                                                          // the rewritten quality is what the vote sees
            const int bin = (int)((uint32_t)(0x4777777377727107ull >> (nb * 4)) & 7u);        // A,C,G,T,N -> 0..4, anything else 7
            if (bin == 7 || (q & 0x80)) s_gflag[x.grp] = 2;
            else {
                uint32_t *t2 = &s_tal[x.ci - c0][bin][0];
                atomicAdd(t2, 1u | ((uint32_t)(sc + p.score_bias) << 6) | ((uint32_t)q << 20)); atomicMax(t2 + 1, (uint32_t)q);
            }
        };
        for (int itb = j0 + tid - lane; itb < j0 + njob; itb += 2 * VB_T) {             // (wave-uniform trips: the side lookup is a wave operation)
            const int sa = s0 + vb_find_wave(s_jpre + s0, s1 - s0, itb, lane, j0 + njob - 1), sb2 = s0 + vb_find_wave(s_jpre + s0, s1 - s0, itb + VB_T, lane, j0 + njob - 1);
            const Item x0 = prep(itb + lane, sa), x1 = prep(itb + lane + VB_T, sb2);
            vote(x0); vote(x1);
        }
#endif
        __syncthreads();
        VB_TICK(6);
        // (d) one lane per column of the round: rule cascade + reference arbitration (group.cpp:394-501)
        for (int cib = c0 + tid - lane; cib < c0 + ncol; cib += VB_T) {
            const int ci = cib + lane;
            const int s = s0 + vb_find_wave(s_cpre + s0, s1 - s0, cib, lane, c0 + ncol - 1);
            if (ci >= c0 + ncol) continue;
            const int col = s_ccol[ci];
            const VSide sd = s_side[s];
            int ref4 = 0;
            if (sd.ref) {                                                               // group.cpp:430-439
                const int ro = sd.o_nc == 1 ? (col < cig_len(sd.o_c0) ? col : -1) : d_ref_offset(b.cigar + b.cigar_off[sd.result], sd.o_nc, col);
                if (ro >= 0 && (int64_t)sd.o_pos + ro < sd.ref_len) ref4 = d_ref_nib_g(sd.ref, (int64_t)sd.o_pos + ro);      // (a GLOBAL load: the pointer comes out of LDS as an integer, and a generic one is a flat_load)
            }
            const int out_base = d_nib(b.seq + s_rd[s & 1][sd.lp0 + sd.tmpl].so, col);
            Tally5 t; t.total = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
#if VB_TPLANE
                const uint32_t *tp_ = &s_tal[0][0][0] + (2 * k) * VB_CCAP + (ci - c0);
                const uint2 v2 = make_uint2(tp_[0], tp_[VB_CCAP]);
#else
                const uint2 v2 = *(const uint2 *)(&s_tal[ci - c0][k][0]);
#endif
                t.cnt[k] = (int)(v2.x & 63u); t.ss[k] = (int)((v2.x >> 6) & 0x3FFFu) - t.cnt[k] * p.score_bias; t.qs[k] = (int)(v2.x >> 20); t.tq[k] = (int)v2.y;
                t.total += t.ss[k];
            }
            const ColOut r = decide_column_packed(t, p, out_base, ref4);
#ifdef VB_COUNT      // what kind of columns reach the full vote (tools: -DVB_PROF -DVB_COUNT): 11 all, 12 one base only, 13 one base only and nothing but the quick accept
            {
                int nb_ = 0; for (int k = 0; k < 5; k++) nb_ += t.cnt[k] > 0;
                atomicAdd(&w.si->prof[11], 1ull); if (nb_ == 1) atomicAdd(&w.si->prof[12], 1ull);
                if (nb_ == 1 && r.base == out_base && r.minc == 0) atomicAdd(&w.si->prof[13], 1ull);
            }
#endif
            s_cq[ci] = (uint8_t)r.qual; s_cb[ci] = (uint8_t)r.base;
            if (r.minc) atomicAdd(&s_side[s].minc, r.minc);
        }
        __syncthreads();
        VB_TICK(7);
        s0 = s1;
        if (s_cpre[s0] >= n_cont) break;
    }
    VB_TICK(8);
    // ---------------------------------------------------------------- P6: results per (group, side)   (group.cpp:528-573)
    // Stores only, and no barrier behind it: what the write-back needs -- "handed on" (s_gflag) and "restore" (mismatchInc > 5) -- every P7 lane
    // derives itself from the side's final minc (side_state below).  The NM patch is DEFERRED: the template's NM tag (two dependent global
    // loads in a phase where the whole block waited for them: 0.33 ms of the kernel for 2 % of its instructions) is looked at by k_group_tail,
    // a thread per group with nothing waiting on it; here only the mismatch delta is noted (NM_DEFER + mismatchInc).
    if (tid < 2 * ng) {
        const int j = tid >> 1, side = tid & 1;
        const uint32_t gi = s_ggi[j];
        const VSide *sd = &s_side[tid];
        if (s_gflag[j] == 2) {                                                          // found out of scope on the way: the whole group goes on, untouched
            if (side == 0) {
                w.gen_flag[2 * gi] = 1; w.gen_flag[2 * gi + 1] = 1;
                const uint32_t np_ = s_gnp[j];
                const unsigned long long o_ = atomicAdd(&w.si->hand_on, (2ull << 32) | (unsigned long long)np_);
                const uint32_t at = (uint32_t)o_, ga = (uint32_t)(o_ >> 32);
                w.gen_list[ga] = 2u * gi; w.gen_list[ga + 1] = 2u * gi + 1u;
                for (uint32_t k = 0; k < np_; k++) w.score_list[at + k] = s_gbeg[j] + k;
            }
        } else if (s_gflag[j] == 0) {
            uint32_t *rp_out = side ? w.rp_right : w.rp_left;
            if (sd->state == VS_ACTIVE) {
                if (sd->minc != 0) w.rp_nm[2 * gi + side] = NM_DEFER + sd->minc;         // -> k_group_tail: NM missing is fatal (quirk Q9), > 5 leaves NM alone, else the patch of group.cpp:569-571
                rp_out[gi] = sd->result;
            } else if (sd->state == VS_FINAL) rp_out[gi] = sd->result;
        }
    }
    // the state of a side as the write-back sees it
    auto side_state = [&](const VSide &sd) -> int {
        if (s_gflag[sd.grp] == 2) return (sd.state == VS_ACTIVE || sd.state == VS_FINAL) ? (int)VS_GEN : (int)sd.state;
        if (s_gflag[sd.grp] == 0 && sd.state == VS_ACTIVE && sd.minc > 5) return (int)VS_RESTORE;
        return (int)sd.state;
    };
    VB_TICK(9);
    // ---------------------------------------------------------------- P7: the templates go back (the only writes to the reads)
    // (a) a restored template (mismatchInc > 5, group.cpp:528-558): seq and qual come back from the backup taken AFTER computeScore
    //     (group.cpp:327-333): the bases stay, the quals are the original ones except where the overlap check rewrote them (pair.cpp:158-159,
    //     quirk Q7).  Read here, written below: the mate may be the other side's template.
    bool any_restore = false;
#pragma unroll
    for (int kk = 0; kk < VB_IPL; kk++) {
        const int it = tid + VB_T * kk;
        if (it < n_items) {
            const int s = item_side[kk];
            const VSide sd = s_side[s];
            if (side_state(sd) == VS_RESTORE) {
                any_restore = true;
                const int chunk = it - (int)sd.item0, c16 = 16 * chunk, nval = min(16, (int)sd.len - c16), side = s & 1;
                const int lp = sd.lp0 + sd.tmpl;
                const VRead *r = &s_rd[side][lp], *mt = &s_rd[side ^ 1][lp];
                const VOv ov = s_ov[lp];
                const int mystart = side ? ov.rs : ov.ls;
                uint32_t qq[4] = {0, 0, 0, 0};
                for (int k = 0; k < nval; k++) {
                    const int c = c16 + k;
                    int q = b.qual[r->qo + c];
                    if ((ov.fl & 2) && (unsigned)(c - mystart) < (unsigned)ov.cmp) {
                        const int mc = c - mystart + (side ? ov.ls : ov.rs);
                        if (d_nib(b.seq + r->so, c) != d_nib(b.seq + mt->so, mc)) q = max(0, q - (int)b.qual[mt->qo + mc]);
                    }
                    qq[k >> 2] |= (uint32_t)q << (8 * (k & 3));
                }
                keep[kk] = make_uint4(qq[0], qq[1], qq[2], qq[3]);
            }
        }
    }
    (void)any_restore;
    __syncthreads();                                // (every read of the originals above precedes every write below; __syncthreads_or kept the packed work-item ids
                                                    //  alive through the whole kernel for a result nobody looked at: one VGPR and, at 64, a spilled pair)
#pragma unroll
    for (int kk = 0; kk < VB_IPL; kk++) {
        const int it = tid + VB_T * kk;
        if (it < n_items) {
            const int s = item_side[kk];
            const VSide sd = s_side[s];
            const int st = side_state(sd);
            if (st != VS_ACTIVE && st != VS_RESTORE) continue;
            const int chunk = it - (int)sd.item0, c16 = 16 * chunk, nval = min(16, (int)sd.len - c16);
            const VRead *r = &s_rd[s & 1][sd.lp0 + sd.tmpl];
            uint8_t *oq = b.qual + r->qo + c16, *os = b.seq + r->so + 8 * chunk;
            uint64_t qlo = (uint64_t)keep[kk].x | ((uint64_t)keep[kk].y << 32), qhi = (uint64_t)keep[kk].z | ((uint64_t)keep[kk].w << 32);
            uint32_t cm = st == VS_ACTIVE ? (s_cmask[s][chunk >> 1] >> (16 * (chunk & 1))) & 0xFFFFu : 0u;
            uint64_t x = 0, x0 = 0; int nbytes = 0;
            if (cm) {
                // the voted columns of this chunk: their place in the side's list = contested columns in front of them
                int ci = (int)s_cpre[s] + (int)s_wbase[s][chunk >> 1];                  // (P5b: columns of the side in front of this 32-column word)
                if (chunk & 1) ci += __popc(s_cmask[s][chunk >> 1] & 0xFFFFu);
                nbytes = min(8, ((int)sd.len + 1) / 2 - 8 * chunk);
                x = ld8_unaligned(os);                                                  // (blobs are readable past a read's last byte)
                if (nbytes < 8) x &= (1ull << (8 * nbytes)) - 1ull;
                x0 = x;
                for (; cm; cm &= cm - 1, ci++) {
                    const int k = __ffs((int)cm) - 1, sh = 8 * (k >> 1) + ((k & 1) ? 0 : 4), qs = 8 * (k & 7);
                    x = (x & ~(0xFull << sh)) | ((uint64_t)(s_cb[ci] & 0xF) << sh);
                    const uint64_t qm = 0xFFull << qs, qv = (uint64_t)s_cq[ci] << qs;
                    if (k & 8) qhi = (qhi & ~qm) | qv; else qlo = (qlo & ~qm) | qv;
                }
            }
            typedef uint64_t u64u __attribute__((aligned(1)));
            typedef uint32_t u32u __attribute__((aligned(1)));
            typedef uint16_t u16u __attribute__((aligned(1)));
            if (nval == 16) { *(u64u *)oq = qlo; *(u64u *)(oq + 8) = qhi; }
            else {                                                                      // the side's last chunk: 8 + 4 + 2 + 1 bytes as needed, never past the read
                uint64_t v = qlo; int o = 0;
                if (nval & 8) { *(u64u *)oq = qlo; v = qhi; o = 8; }
                if (nval & 4) { *(u32u *)(oq + o) = (uint32_t)v; v >>= 32; o += 4; }
                if (nval & 2) { *(u16u *)(oq + o) = (uint16_t)v; v >>= 16; o += 2; }
                if (nval & 1) oq[o] = (uint8_t)v;
            }
            if (x != x0) {                                                               // (the lane owns the chunk's bytes: one store when all eight are the read's)
                if (nbytes == 8) *(u64u *)os = x;
                else for (int k = 0; k < nbytes; k++) if ((uint8_t)(x >> (8 * k)) != (uint8_t)(x0 >> (8 * k))) os[k] = (uint8_t)(x >> (8 * k));
            }
        }
    }
    VB_TICK(10);
}
