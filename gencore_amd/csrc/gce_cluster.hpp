// gce_cluster.hpp — cluster formation: Gencore::addToProperCluster, src/gencore.cpp:295-434.
//
//   k_cluster      THE CLUSTERING SCAN.  One pass over the 32-byte key records, SB_READS (1024) reads per block: class, cluster key
//                  (gencore.cpp:295-312), the sortedness check (gencore.cpp:233-241), and block-level aggregation in LDS -- the first
//                  read of every distinct key of the block is its LEADER, the others draw a block-local rank from the leader's counter.
//                  Output per read: (leader record, rank in the leader's run) = 8 bytes.  The kernel touches NO stream-global state:
//                  no tick, no flush event, no bucket table, no device-scope atomic (round 2's scan spent 8 of its 16.7 us per block in
//                  two fabric atomics per (cluster, block), issued by ~80 of the block's 512 lanes while the others waited).
//                  Why it can: a read's cluster INSTANCE depends on the flush events only if some event IN FRONT of
//                  the read takes its key, i.e. tid == T && left < P && right < P for an event read at (T, P) <= (tid, pos).  With
//                  right >= pos that is impossible whatever the events are, so every such read ("fast": all proper pairs) belongs to the
//                  instance its key implies, and reads with equal keys can be merged blindly.  The others ("odd": right < pos --
//                  cross-contig keys, inconsistent isize) become leaders of their own and are decided by k_leaders.
//   k_blk_scan     exclusive scan of the blocks' clustered counts -> tick of every leader (gencore.cpp:319-320)
//   k_events       the reads on which the periodic flush fires (gencore.cpp:319-322)
//   k_leaders      one lane per LEADER (~0.15 per read): instance from the flush events (closed form), then the bucket
//                  table -- one CAS + one 64-bit add per (cluster, block) as before, but issued by full waves of a latency-tolerant kernel
//   k_num_*        clusters numbered in the order of their claiming leaders, exclusive scan of (1 << 32 | reads) -> (cluster id, first
//                  member slot); runs over the leaders, not over the reads
//   k_scatter      members[] (CSR) from (leader, rank): the second and last pass over per-read data; the claiming leaders wipe their buckets
//                  on the way (the table is all-zero again when the step ends: no 16 B x 1.25 N memset per step)
//   k_describe     NOT part of cluster formation: per read the 32-byte ReadDesc, BamUtil::getUMI (bamutil.cpp:23-112) and the
//                  pre-Stats counters (stats.cpp:101-121) that pairing and the vote consume; independent of everything above
#pragma once

#ifndef SB_T
#define SB_T 512                         // threads per scan block (round 5: 256 -> 512, scan blocks of 1024 reads: a capture panel spreads a cluster's reads over ~540 reads of the stream,
#endif                                   // so that a cluster met 2.06 blocks of 512 reads = leader runs = read-modify-writes of k_leaders, and meets 1.5 of 1024)
#ifndef SB_U
#define SB_U 2                           // reads per thread
#endif
#define SB_READS (SB_T * SB_U)           // reads per scan block
#define SB_LDS_SLOTS (2 * SB_READS)       // LDS hash slots for the <= SB_READS distinct keys of a block
#define SB_OFF_BITS (SB_READS == 1024 ? 10 : 9)        // a read's place in its scan block
static_assert(SB_READS == 512 || SB_READS == 1024, "LeadRec.info packs a place and a count of SB_READS");

// What a leader hands to k_leaders.  info: bits 0..SB_OFF_BITS-1 its offset inside the scan block; the 11 bits above the clustered reads of the block in
// front of it (tick = block base + that + 1); bit 22 "odd" (an earlier flush event may have taken the key: run the event test).
// kw: the packed key word (d_pack_key) or 0 when the key does not fit it (k_leaders derives the key from the leader's key record).
struct __attribute__((aligned(16))) LeadRec { unsigned long long kw; uint32_t info, runlen; };
static_assert(sizeof(LeadRec) == 16, "LeadRec must stay 16 bytes");
#define LI_ODD (1u << 22)
struct __attribute__((aligned(8))) BlkHdr { uint32_t n_lead, n_clu; };
struct __attribute__((aligned(8))) LeadOut { uint32_t h, rb; };     // bucket; first in-cluster rank of the run | LO_OWNER
#define LO_OWNER 0x80000000u

// thr_mode of an instance (see DESIGN.md "flush rule in closed form")
__device__ __forceinline__ uint32_t d_thr_mode(uint32_t ikey, const StreamInfo *si, const DevParams &p) {
    uint32_t inst = ikey & 0x7FFFFFFFu, seg_b = ikey >> 31;
    if (!seg_b) {
        if ((int)(inst + 1) <= si->n_events_a) return THR_PROPER;
        if (si->first_unmapped != NONE32) return THR_UNPROPER;
        return p.trailing_flush ? THR_PROPER : THR_UNPROPER;
    }
    return ((int)(inst + 1) <= si->n_events) ? THR_PROPER : THR_NEVER;
}

// The bucket table: 16 bytes per bucket, the cluster's whole identity in the CAS word itself so that nobody ever has to WAIT for a
// claimer to publish something.
//   key   bit 63 = occupied (0 = empty), bit 62 = EXOTIC
//         normal : (x div T) | (right - left + 1) | reads of the cluster so far, in the cluster's HOME bucket x mod T only (x = genome-linear
//                  left * 8 + way): identity and count in ONE word, see k_leaders.  The instance is NOT part of it: a read whose instance
//                  is the one its key implies (the overwhelming case) shares it with every other such read of the key.
//         exotic : instance (29 bits) | segment << 29 in bits 32..61, the CLAIMING READ's index in bits 0..31 -- for everything else:
//                  cross-contig keys (negative right), fields that overflow the packing, reads that arrive after their key was
//                  flushed (instance = own epoch), reads behind an unmapped read.  A follower compares the upper half, then the
//                  cluster key of the claiming read (one dependent load of its key record; rare).
//   ic    exotic entries only.  Low half: reads of the cluster so far -- one 64-bit atomicAdd per run gives the in-cluster ranks; high
//         half: instance | segment << 31 of the cluster (the UMI threshold of quirk Q1 follows from it), added in by the claimer
struct __attribute__((aligned(16))) TabEntry { unsigned long long key; unsigned long long ic; };   // ic = ikey << 32 | count
static_assert(sizeof(TabEntry) == 16, "TabEntry must stay 16 bytes");
#define TAB_OCC (1ull << 63)
#define TAB_EXO (1ull << 62)

// Bucket of a cluster key.  The stream is coordinate sorted, so consecutive leaders carry neighbouring `left` values: a
// LOCALITY-PRESERVING bucket index (genome-linear left, TAB_WAYS buckets per position, the way from right / instance) makes the
// table accesses a sliding window instead of 64-byte random HBM touches.  Capture panels stack hundreds of clusters on a few hundred
// positions: 8 ways keep the local load low there.  Collisions leave the neighbourhood through a hashed second probe.
#define TAB_WAYS 8
__device__ __forceinline__ uint64_t d_tab_index(const ClusterKey &k, uint32_t ikey, const DevParams &p) {
    uint64_t g = (k.tid >= 0 && k.tid < p.n_targets && p.target_cum) ? p.target_cum[k.tid] : (uint64_t)(uint32_t)k.tid * 0x9E3779B97F4A7C15ull;
    uint64_t m = ((uint64_t)k.right * 0x165667B19E3779F9ull) ^ ((uint64_t)ikey * 0xD6E8FEB86659FD93ull);
    m ^= m >> 29;
    return (((g + (uint64_t)(uint32_t)k.left) & 0x0000FFFFFFFFFFFFull) * TAB_WAYS) | (m & (TAB_WAYS - 1));
}
// the normal key word, or 0 when a field does not fit
__device__ __forceinline__ unsigned long long d_pack_key(const ClusterKey &k, const DevParams &p) {
    const long long delta1 = k.right - (long long)k.left + 1;                     // |isize| for a nearby pair
    const int bd = 62 - p.key_bt - p.key_bl;
    if (k.tid >= 0 && ((uint64_t)(uint32_t)k.tid >> p.key_bt) == 0 && k.left >= 0 && ((uint64_t)(uint32_t)k.left >> p.key_bl) == 0 &&
        delta1 >= 0 && bd > 0 && ((uint64_t)delta1 >> bd) == 0)
        return TAB_OCC | (uint64_t)(uint32_t)k.tid | ((uint64_t)(uint32_t)k.left << p.key_bt) | ((uint64_t)delta1 << (p.key_bt + p.key_bl));
    return 0ull;
}
__device__ __forceinline__ ClusterKey d_unpack_key(unsigned long long kw, const DevParams &p) {
    ClusterKey k;
    k.tid = (int32_t)(kw & ((1ull << p.key_bt) - 1ull));
    k.left = (int32_t)((kw >> p.key_bt) & ((1ull << p.key_bl) - 1ull));
    const long long delta1 = (long long)((kw & ~(TAB_OCC | TAB_EXO)) >> (p.key_bt + p.key_bl));
    k.right = (long long)k.left + delta1 - 1;
    return k;
}
__device__ __forceinline__ unsigned long long d_exotic_key(uint32_t ikey, uint32_t read) {
    return TAB_OCC | TAB_EXO | ((uint64_t)((ikey & 0x1FFFFFFFu) | ((ikey >> 31) << 29)) << 32) | read;
}

// ===================================================================================================== the clustering scan
#ifdef CL_PROF
#define CL_TICK(k) do { if (threadIdx.x == 0 && (blockIdx.x & 31) == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&w.si->prof[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define CL_TICK(k) do { } while (0)
#endif
__global__ __launch_bounds__(SB_T, 2048 / SB_T) void k_cluster(DevBatch b, DevParams p, Work w) {
    __shared__ uint32_t s_slot[SB_LDS_SLOTS];
    __shared__ unsigned long long s_kw[SB_READS];
    __shared__ uint32_t s_cnt[SB_READS];
    __shared__ uint16_t s_num[SB_READS];
    __shared__ uint32_t s_wcnt[SB_U][SB_T / 64];
    __shared__ uint32_t s_nlead;
    __shared__ uint32_t s_unm[SB_T / 64];                                                  // per wave: an unmapped read among its reads
    const int lane = lane_id(), wv = threadIdx.x >> 6;
#ifdef CL_PROF
    unsigned long long t_prev_ = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0) atomicAdd(&w.si->prof[15], 1ull);
#endif
    int64_t idx[SB_U]; gce_core k[SB_U]; int2 pv[SB_U];
#pragma unroll
    for (int u = 0; u < SB_U; u++) {
        idx[u] = ((int64_t)blockIdx.x * SB_U + u) * SB_T + threadIdx.x;
        union { gce_core c; uint4 q[2]; } t;                     // the 32-byte key record as two 16-byte loads (core[] is 16-byte aligned)
        t.q[0] = make_uint4(0, 0, 0, 0); t.q[1] = t.q[0];
        pv[u] = make_int2(-1, -1);
        if (idx[u] < b.n) {
            const uint4 *src = reinterpret_cast<const uint4 *>(b.core + idx[u]); t.q[0] = src[0]; t.q[1] = src[1];
            if (idx[u] > 0) pv[u] = *reinterpret_cast<const int2 *>(b.core + idx[u] - 1);      // (tid, pos) of the read in front: the neighbour's sector
        }
        k[u] = t.c;
    }
    for (int q = threadIdx.x; q < SB_LDS_SLOTS; q += SB_T) s_slot[q] = 0;
    if (threadIdx.x == 0) s_nlead = 0;
    bool cl[SB_U], odd[SB_U], unm = false; unsigned long long m[SB_U], kw[SB_U];
#pragma unroll
    for (int u = 0; u < SB_U; u++) {
        const bool in = idx[u] < b.n;
        const uint8_t c = in ? d_classify(k[u]) : (uint8_t)CLS_DROP;
        cl[u] = c == CLS_CLUSTERED;
        bool unm_u = false;
        if (in) {
            if (idx[u] > 0 && (k[u].tid < pv[u].x || (k[u].tid == pv[u].x && k[u].pos < pv[u].y)) && k[u].tid >= 0 && k[u].pos >= 0)
                raise_error(w.si, GCE_ERR_UNSORTED, (uint32_t)idx[u]);                            // gencore.cpp:233-241
            if (k[u].tid < 0 || k[u].pos < 0) unm_u = true;                                       // gencore.cpp:255-262 (rare: the tail of a file)
            if (c == CLS_BYPASS) w.out_flag[idx[u]] = 2;                                          // mate unmapped: written as it is (gencore.cpp:307-309)
        }
        if (__any(unm_u)) {                                                                       // one guarded atomic per wave: a file's unmapped tail is millions of reads
            unm = true;
            const unsigned int v = (unsigned int)wave_min((int)((unm_u ? (unsigned int)idx[u] : NONE32) ^ 0x80000000u)) ^ 0x80000000u;   // unsigned min via signed flip
            if (lane == 0 && v < *(volatile unsigned int *)&w.si->first_unmapped) atomicMin(&w.si->first_unmapped, v);
        }
        m[u] = __ballot(cl[u]);
        if (lane == 0) s_wcnt[u][wv] = (uint32_t)__popcll(m[u]);
        kw[u] = 0ull; odd[u] = false;
        if (cl[u]) {
            const ClusterKey key = d_key(k[u], p);
            odd[u] = key.right < (long long)k[u].pos;
            kw[u] = d_pack_key(key, p);
        }
        const int id = u * SB_T + threadIdx.x;
        s_kw[id] = kw[u]; s_cnt[id] = 0u;
    }
    // an unmapped read in the block: the reads behind it belong to the stream's second segment (gencore.cpp:255-262 ran
    // finishConsensus in between) and must not merge with the ones in front -- such a block (the tail of a file, as a rule: no
    // clustered reads at all) does not aggregate, every read is its own leader and the bucket table sorts it out
    // (every wave leaves its own word and the barrier the block needs anyway follows: __syncthreads_or is a work-group reduction with barriers of its own)
    {
        const unsigned long long um = __ballot(unm);
        if (lane == 0) s_unm[wv] = um != 0ull;
    }
    __syncthreads();
    int any_unm = 0;
#pragma unroll
    for (int q = 0; q < SB_T / 64; q++) any_unm |= (int)s_unm[q];
    CL_TICK(0);
    int leader[SB_U]; uint32_t lrank[SB_U];
#pragma unroll
    for (int u = 0; u < SB_U; u++) {
        leader[u] = -1; lrank[u] = 0;
        if (cl[u]) {
            const int id = u * SB_T + threadIdx.x;
            leader[u] = id;
            if (kw[u] != 0ull && !odd[u] && !any_unm) {
                uint32_t hs = ((uint32_t)kw[u] ^ (uint32_t)(kw[u] >> 29)) * 0x9E3779B1u;
                hs = (hs ^ (hs >> 15)) & (SB_LDS_SLOTS - 1);
                for (;;) {                                                                  // (no waiting: a claimed slot's key was stored before the barrier)
                    const uint32_t old = atomicCAS(&s_slot[hs], 0u, (uint32_t)id + 1u);
                    if (old == 0u) break;
                    const int L = (int)old - 1;
                    if (s_kw[L] == kw[u]) { leader[u] = L; break; }
                    hs = (hs + 1) & (SB_LDS_SLOTS - 1);
                }
            }
            lrank[u] = atomicAdd(&s_cnt[leader[u]], 1u);
            if (leader[u] == id) s_num[id] = (uint16_t)atomicAdd(&s_nlead, 1u);
        }
    }
    __syncthreads();
    CL_TICK(1);
    uint32_t front = 0;                                                                     // clustered reads of the block in front of this thread's read u
#pragma unroll
    for (int u = 0; u < SB_U; u++) {
        const int id = u * SB_T + threadIdx.x;
        uint32_t inblock = front + (uint32_t)lanes_below(m[u]);
#pragma unroll
        for (int q = 0; q < SB_T / 64; q++) { const uint32_t c = s_wcnt[u][q]; inblock += q < wv ? c : 0u; front += c; }
        if (idx[u] < b.n) {
            if (cl[u]) {
                w.slot[idx[u]] = (uint32_t)s_num[leader[u]] | lrank[u] << 16;                 // (leader of the block, rank in its run): SB_OFF_BITS + SB_OFF_BITS bits (10 + 10 at SB_READS = 1024), the block is idx / SB_READS
                if (leader[u] == id) {
                    union { LeadRec r; uint4 q; } o;
                    o.r.kw = kw[u]; o.r.info = (uint32_t)id | inblock << SB_OFF_BITS | (odd[u] ? LI_ODD : 0u); o.r.runlen = s_cnt[id];
                    *reinterpret_cast<uint4 *>(w.lrec + (size_t)blockIdx.x * SB_READS + s_num[id]) = o.q;
                }
            } else w.slot[idx[u]] = NONE32;
        }
    }
    if (threadIdx.x == 0) { BlkHdr h; h.n_lead = s_nlead; h.n_clu = front; w.bhdr[blockIdx.x] = h; }
    CL_TICK(2);
}

// single-block exclusive scan of the blocks' clustered counts -> blk_base; the stream's totals.  Every WAVE owns one contiguous range
// of blocks and walks it twice, 256 blocks (four per lane, coalesced) per trip: sums first, one barrier for the sixteen wave totals,
// then the prefixes with the carry in a register -- no block barrier inside the loops (rounds of 8192 blocks with a barrier each
// were 34 us for 39 k blocks).
__global__ __launch_bounds__(1024) void k_blk_scan(Work w, DevParams p) {
    __shared__ unsigned int s_w[16], s_l[16];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int64_t per_w = ((w.n_sblk + 15) / 16 + 255) / 256 * 256, w0 = per_w * wv, w1 = min(w0 + per_w, w.n_sblk);
    auto load4 = [&](int64_t i, unsigned int (&c)[4], unsigned int &nl) {
        if (i + 3 < w1) {
            const uint4 q0 = reinterpret_cast<const uint4 *>(w.bhdr + i)[0], q1 = reinterpret_cast<const uint4 *>(w.bhdr + i)[1];     // (i is a multiple of 4: 32-byte aligned)
            c[0] = q0.y; c[1] = q0.w; c[2] = q1.y; c[3] = q1.w; nl += q0.x + q0.z + q1.x + q1.z;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { c[k] = 0; if (i + k < w1) { const BlkHdr h = w.bhdr[i + k]; c[k] = h.n_clu; nl += h.n_lead; } }
        }
    };
    unsigned int v = 0, nl = 0;
    for (int64_t i = w0 + 4 * lane; i < w1; i += 256) { unsigned int c[4]; load4(i, c, nl); v += c[0] + c[1] + c[2] + c[3]; }
    v = (unsigned int)wave_sum((int)v); nl = (unsigned int)wave_sum((int)nl);
    if (lane == 0) { s_w[wv] = v; s_l[wv] = nl; }
    __syncthreads();
    unsigned int carry = 0, tot = 0; unsigned long long leaders = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const unsigned int t = s_w[k]; tot += t; carry += k < wv ? t : 0u; leaders += s_l[k]; }
    for (int64_t i0 = w0; i0 < w1; i0 += 256) {                                   // (wave-uniform trips: the scan is a wave operation)
        const int64_t i = i0 + 4 * lane;
        unsigned int c[4], dummy = 0; load4(i, c, dummy);
        const unsigned int sum = c[0] + c[1] + c[2] + c[3];
        const unsigned int x = (unsigned int)wave_scan_incl((int)sum);
        unsigned int run = carry + x - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (i + k < w1) w.blk_base[i + k] = run; run += c[k]; }
        carry += (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
    }
    if (threadIdx.x == 0) {
        unsigned long long total = tot;
        w.si->n_clustered = total;
        w.si->n_leaders = leaders;
        long long per_ = p.period;
        long long e = (p.tick_offset + (long long)total) / per_ - p.tick_offset / per_;
        w.si->n_events = (int)(e < w.max_events ? e : w.max_events);
        if (e > w.max_events) raise_error(w.si, GCE_ERR_INVALID, 0);
    }
}

// one WAVE per flush event: find the read on which tick % period == 0 (gencore.cpp:319-322) -- the scan block by a 64-ary search over the
// block bases, the read inside it by ballots over the classes of the block's SB_READS key records
#define EV_T 1024                                                     // 16 events per block: one global atomic per block for the segment count
__global__ __launch_bounds__(EV_T) void k_events(DevBatch b, DevParams p, Work w) {
    __shared__ int s_a;
    if (threadIdx.x == 0) s_a = 0;
    __syncthreads();
    const int j = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = lane_id();      // event j+1
    const int E = w.si->n_events;
    if (j < E) {
        const long long per = p.period;
        const long long gt = (p.tick_offset / per + (j + 1)) * per;      // global tick of this event
        const unsigned int t = (unsigned int)(gt - p.tick_offset);       // local inclusive count, >= 1
        int64_t lo = 0, hi = w.n_sblk - 1;                               // last block with blk_base < t (blk_base[0] = 0 < t):
        while (lo < hi) {                                                // 64-ary search, three dependent round trips instead of seventeen
            const int64_t step = (hi - lo + 63) / 64, probe = lo + (int64_t)(lane + 1) * step;
            const bool below = probe <= hi && w.blk_base[probe] < t;     // true for a prefix of the lanes (the bases do not decrease)
            const int k = __popcll(__ballot(below));
            hi = min(hi, lo + (int64_t)(k + 1) * step - 1);
            lo = lo + (int64_t)k * step;
        }
        unsigned int need = t - w.blk_base[lo];                          // the need-th clustered read of the block
        const int64_t i0 = lo * SB_READS, end = min(b.n, i0 + SB_READS);
        int64_t i = end;
        bool clq[SB_READS / 64];                                         // (all of the block's classes in one round trip)
#pragma unroll
        for (int q = 0; q < SB_READS / 64; q++) {
            const int64_t idx = i0 + 64 * q + lane;
            clq[q] = idx < end && w.slot[idx] != NONE32;                    // k_cluster left NONE32 for every read that is not clustered: 4 bytes per read instead of its 32-byte key record
        }
#pragma unroll
        for (int q = 0; q < SB_READS / 64; q++) {
            const bool cl = clq[q];
            const unsigned long long m = __ballot(cl);
            const unsigned int cnt = (unsigned int)__popcll(m);
            if (need <= cnt) {
                const unsigned long long hit = __ballot(cl && (unsigned int)__popcll(m & ((1ull << lane) - 1ull)) + 1u == need);
                i = i0 + 64 * q + (__ffsll((long long)hit) - 1);
                break;
            }
            need -= cnt;
        }
        if (lane == 0) {
            w.ev_read[j] = (uint32_t)i;
            w.ev_tid[j] = b.core[i].tid;
            w.ev_pos[j] = b.core[i].pos;
            if ((uint32_t)i < w.si->first_unmapped) atomicAdd(&s_a, 1);  // (2000 adds to one global word were 24 of the kernel's 47 us)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_a) atomicAdd(&w.si->n_events_a, s_a);
}

// ===================================================================================================== leaders -> bucket table
// One wave per scan block, one lane per leader of it.  Instance of a leader: events before it (own epoch e) vs. the first event whose
// walk takes its key (gencore.cpp:333-354: tid < T || (tid == T && left < P && right < P), monotone in the event index).
__global__ __launch_bounds__(256) void k_leaders(DevBatch b, DevParams p, Work w) {
    const int lane = lane_id();
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= w.n_sblk) return;
    const uint32_t nl = w.bhdr[blk].n_lead;
    if (nl == 0) return;
    const StreamInfo *si = w.si;
    const unsigned int first_unm = si->first_unmapped;
    const int n_ev_a = si->n_events_a, n_ev = si->n_events;
    const long long per = p.period;
    const int ev_last = max(n_ev - 1, 0);
    // ticks in front of the block, from the epoch boundary: epoch E and remainder r (wave-uniform)
    unsigned int E = 0, r = 0;
    if (!b.tick) {
        const unsigned long long t0 = (unsigned long long)(unsigned int)p.tick_rem0 + w.blk_base[blk];
        const unsigned int uper = (unsigned int)per;
        if ((t0 >> 32) == 0) { E = (unsigned int)t0 / uper; r = (unsigned int)t0 - E * uper; }
        else { E = (unsigned int)(t0 / uper); r = (unsigned int)(t0 - (unsigned long long)E * uper); }
    }
    for (uint32_t kq = (uint32_t)lane; kq < nl; kq += 64) {
        const size_t LR = (size_t)blk * SB_READS + kq;
        union { LeadRec r; uint4 q; } in; in.q = *reinterpret_cast<const uint4 *>(w.lrec + LR);
        const uint32_t off = in.r.info & ((1u << SB_OFF_BITS) - 1u), rk = (in.r.info >> SB_OFF_BITS) & 0x7FFu;
        const bool odd = (in.r.info & LI_ODD) != 0;
        const uint32_t idx = (uint32_t)(blk * SB_READS + off);
        ClusterKey key;
        if (in.r.kw) key = d_unpack_key(in.r.kw, p);
        else {
            union { gce_core c; uint4 q[2]; } t; const uint4 *src = reinterpret_cast<const uint4 *>(b.core + idx); t.q[0] = src[0]; t.q[1] = src[1];
            key = d_key(t.c, p);
        }
        // the reference's `tick` after ++ (gencore.cpp:319-320) -> flush events before this read, (tick - 1) / period
        int e;
        if (b.tick) e = (int)(((long long)b.tick[idx] - 1) / per - p.tick_epoch0);
        else { const unsigned int x = r + rk; e = (int)(E + x / (unsigned int)per); }
        const bool seg_b = (first_unm != NONE32) && (idx > first_unm);
        const int lo = seg_b ? n_ev_a : 0, hi = seg_b ? n_ev : n_ev_a;                      // events [lo, hi) 0-based
        const int g = min(max(e, lo), hi);
        auto takes = [&](int T, int P) { return key.tid < T || (key.tid == T && key.left < P && key.right < (long long)P); };
        // the answer is almost always the read's own epoch or the next one
        const int j0 = min(max(g - 1, 0), ev_last), j1 = min(g, ev_last), j2 = min(g + 1, ev_last);
        const int T0 = w.ev_tid[j0], P0 = w.ev_pos[j0], T1 = w.ev_tid[j1], P1 = w.ev_pos[j1], T2 = w.ev_tid[j2], P2 = w.ev_pos[j2];
        // NORMAL clusters (implied instance, first segment, key inside the header's contigs): the bucket word in the cluster's HOME
        // bucket identifies it by itself.  x = genome-linear left * 8 + way(right) is injective in (tid, left) up to the way, home = x mod T,
        // so (x div T, right - left + 1) is all that is left of the key: word = OCC | x div T | delta1 | reads so far (field widths from the
        // stream's size, DevParams.nw_*).  ONE read-modify-write per leader run: a CAS from 0 claims the bucket, a CAS from (word) to
        // (word + run length) joins it and returns the run's first rank in the same trip; no second word, no look at anybody's record.
        // A cluster that finds its home taken by another identity goes the exotic way (below) -- every one of its runs sees the same, the
        // first claimer of a bucket stays.  tools/mb/atomics.hip: the L2 atomic unit bounds this kernel (17 G/s on lines it has to fetch,
        // 32 G/s on lines a load brought in, whatever the scope), so the first look is a load, issued for fast leaders behind the event
        // loads (answers come back in order) while the instance is still being worked out.
        const int cb = p.nw_cb, bd = p.nw_bd;
        const long long delta1 = key.right - (long long)key.left + 1;
        bool fits = false; uint64_t h = 0; unsigned long long id_w = 0, cur = 0;
        if (p.nw_ok && in.r.kw != 0ull && !seg_b && key.tid < p.n_targets) {
            uint64_t q;
            d_divmod(d_tab_index(key, 0u, p), w.tsize, w.tinv, q, h);
            fits = (q >> (62 - cb - bd)) == 0 && ((uint64_t)delta1 >> bd) == 0;
            id_w = TAB_OCC | (q << (cb + bd)) | ((uint64_t)delta1 << cb);
        }
        const bool early = fits && !odd;
        if (early) cur = __hip_atomic_load(&w.tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int a, z;
        if (odd && g > lo && takes(T0, P0)) { a = lo; z = g - 1; }                          // already flushable before its own epoch
        else if (g >= hi || takes(T1, P1)) a = z = g;
        else if (g + 1 >= hi || takes(T2, P2)) a = z = g + 1;
        else { a = g + 2; z = hi; }
        while (a < z) {
            const int mid = (a + z) >> 1;
            if (takes(w.ev_tid[mid], w.ev_pos[mid])) z = mid; else a = mid + 1;
        }
        const bool implied = e <= a;                                                        // (f = a + 1, 1-based: the instance every early read of the key gets)
        const uint32_t ikey = (seg_b ? 0x80000000u : 0u) | (uint32_t)max(e, a);
        const bool normal = fits && implied;
        bool owner = false, exotic = !normal, first_miss = true; uint32_t rbase = 0;
        auto next_probe = [&]() {
            // collision: leave the neighbourhood.  A deep amplicon stacks thousands of clusters on a few hundred positions; their
            // buckets are full, and walking on linearly would crawl through the whole pile.  The probe sequence continues at a
            // hashed place of the table (load there: a few percent), linearly from then on.
            if (first_miss) {
                uint64_t x = ((uint64_t)(uint32_t)key.tid * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)key.left * 0xC2B2AE3D27D4EB4Full) ^ ((uint64_t)key.right * 0x165667B19E3779F9ull) ^ ((uint64_t)ikey * 0xD6E8FEB86659FD93ull);
                x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
                h = d_bucket(x >> 14, w.tsize, w.tinv);
                first_miss = false;
            } else h = h + 1 == w.tsize ? 0 : h + 1;
        };
        if (normal) {
            const unsigned long long cmask = (1ull << cb) - 1ull;
            if (!early) cur = __hip_atomic_load(&w.tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (cur == 0ull) {
                    const unsigned long long o = atomicCAS(&w.tab[h].key, 0ull, id_w | in.r.runlen);
                    if (o == 0ull) { owner = true; break; }
                    cur = o;
                }
                if ((cur >> cb) != (id_w >> cb)) { exotic = true; break; }                  // (an exotic word differs in bit 62)
                if ((cur & cmask) + in.r.runlen > cmask) { raise_error(w.si, GCE_ERR_INVALID, idx); break; }   // cannot happen: the field holds the stream's read count
                const unsigned long long o = atomicCAS(&w.tab[h].key, cur, cur + in.r.runlen);
                if (o == cur) { rbase = (uint32_t)(cur & cmask); break; }
                cur = o;                                                                    // (the count moved on; the identity bits never change)
            }
            if (owner) w.lrec[LR].runlen = ikey;                                            // the cluster's instance, for k_num_reduce (the run length is spent; info tells the phases apart)
            if (exotic) next_probe();                                                       // home is somebody else's: on to the hashed probe
        } else h = d_bucket(d_tab_index(key, ikey, p), w.tsize, w.tinv);
        if (exotic) {
            // EXOTIC: instance + claiming read in the word, the count in the entry's second word (CAS, then a 64-bit add)
            const unsigned long long tk = d_exotic_key(ikey, idx);
            if ((ikey & 0x7FFFFFFFu) >= (1u << 29)) raise_error(w.si, GCE_ERR_INVALID, idx);                // > 2^29 flush events
            for (;;) {
                const unsigned long long c = atomicCAS(&w.tab[h].key, 0ull, tk);
                if (c == 0ull) { owner = true; break; }
                bool mine = false;
                if ((c >> 32) == (tk >> 32)) { const ClusterKey ok = d_key(b.core[(uint32_t)c], p); mine = ok.tid == key.tid && ok.left == key.left && ok.right == key.right; }
                if (mine) break;
                next_probe();
            }
            rbase = (uint32_t)atomicAdd(&w.tab[h].ic, (unsigned long long)in.r.runlen | (owner ? (unsigned long long)ikey << 32 : 0ull));
        }
        LeadOut o; o.h = (uint32_t)h; o.rb = rbase | (owner ? LO_OWNER : 0u);
        w.lout[LR] = o;
    }
}

// ===================================================================================================== cluster list + member lists
// Every cluster has exactly one claiming leader (LO_OWNER): the clusters are numbered in the order of those leaders, and an
// exclusive scan of (1 << 32 | reads of the cluster) over them gives (cluster id, first member slot).  One wave per scan block.
#define SCAN_TILE 2048
__global__ __launch_bounds__(256) void k_num_reduce(Work w, int p_cb) {
    const int lane = lane_id();
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool on = blk < w.n_sblk;                                                         // (no early return: the block meets at a barrier below)
    const uint32_t nl = on ? w.bhdr[blk].n_lead : 0u;
    uint64_t v = 0;
    for (uint32_t kq = (uint32_t)lane; kq < nl; kq += 64) {
        const size_t LR = (size_t)blk * SB_READS + kq;
        const LeadOut o = w.lout[LR];
        if (o.rb & LO_OWNER) {
            const TabEntry t = w.tab[o.h];
            const unsigned long long ic = (t.key & TAB_EXO) ? t.ic : (((unsigned long long)w.lrec[LR].runlen << 32) | (t.key & ((1ull << p_cb) - 1ull)));   // instance << 32 | reads
            w.lrec[LR].kw = ic;                                                             // (the leader record is spent: it keeps that for k_num_apply)
            v += (1ull << 32) | (uint32_t)ic;
        }
    }
    v = (uint64_t)wave_sum64((long long)v);
    // one partial per BLOCK of four scan blocks goes through k_scan_partials (a single workgroup: 33 us for 39 k partials, 10 us for a quarter of
    // them); the waves' own totals stay unscanned behind them (scan_part[nb4 + blk]) and k_num_apply adds up those of the waves in front of it
    __shared__ uint64_t s_v[4];
    if (lane == 0) { s_v[threadIdx.x >> 6] = v; if (on) w.scan_part[gridDim.x + blk] = v; }
    __syncthreads();
    if (threadIdx.x == 0) w.scan_part[blockIdx.x] = s_v[0] + s_v[1] + s_v[2] + s_v[3];
}
// element(h) = (count>0) << 32 | count ; used by the small per-cluster scans below
__device__ __forceinline__ uint64_t tab_elem(const uint32_t *cnt, uint64_t h, uint64_t n) {
    if (h >= n) return 0;
    uint32_t c = cnt[h];
    return ((uint64_t)(c > 0) << 32) | c;
}
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *cnt, uint64_t n, uint64_t *part) {
    __shared__ uint64_t s[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE, v = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) v += tab_elem(cnt, base + k * 256 + threadIdx.x, n);
    v = (uint64_t)wave_sum64((long long)v);
    if (lane_id() == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// exclusive scan of the tile totals, one block: eight consecutive values per thread and step, one barrier per 8192 values (the wave
// totals are double-buffered and every thread adds them up for the carry itself)
__global__ __launch_bounds__(1024) void k_scan_partials(uint64_t *part, uint64_t nparts, unsigned long long *total_hi, unsigned long long *total_lo) {
    __shared__ uint64_t s_w[2][16];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t carry = 0;
    int it = 0;
    for (uint64_t base = 0; base < nparts; base += 8192, it ^= 1) {
        const uint64_t i0 = base + 8 * (uint64_t)threadIdx.x;
        uint64_t c[8], v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { c[k] = i0 + k < nparts ? part[i0 + k] : 0; v += c[k]; }
        uint64_t x = v;
        for (int o = 1; o < 64; o <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, o); if (lane >= o) x += t; }
        if (lane == 63) s_w[it][wv] = x;
        __syncthreads();
        uint64_t woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { const uint64_t t = s_w[it][k]; tot += t; woff += k < wv ? t : 0ull; }
        uint64_t run = carry + woff + x - v;
#pragma unroll
        for (int k = 0; k < 8; k++) { if (i0 + k < nparts) part[i0 + k] = run; run += c[k]; }
        carry += tot;
    }
    if (threadIdx.x == 0) { *total_hi = carry >> 32; if (total_lo) *total_lo = carry & 0xFFFFFFFFull; }
}
__global__ __launch_bounds__(256) void k_num_apply(Work w) {
    const int lane = lane_id();
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= w.n_sblk) return;
    const uint32_t nl = w.bhdr[blk].n_lead;
    uint64_t carry = w.scan_part[blockIdx.x];                                   // (scanned: the blocks in front) + the waves in front of this one
    for (int q = 0; q < (int)(threadIdx.x >> 6); q++) carry += w.scan_part[gridDim.x + blk - (threadIdx.x >> 6) + q];
    for (uint32_t k0 = 0; k0 < nl; k0 += 64) {                                  // (wave-uniform trips: the scan is a wave operation)
        const uint32_t kq = k0 + (uint32_t)lane;
        uint64_t v = 0, ic = 0; uint32_t h = 0;
        if (kq < nl) {
            const size_t LR = (size_t)blk * SB_READS + kq;
            const LeadOut o = w.lout[LR];
            if (o.rb & LO_OWNER) { h = o.h; ic = w.lrec[LR].kw; v = (1ull << 32) | (uint32_t)ic; }
        }
        uint64_t x = v;
        for (int o = 1; o < 64; o <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, o); if (lane >= o) x += t; }
        if (v) {
            const uint64_t ex = carry + x - v;
            const uint32_t cid = (uint32_t)(ex >> 32);
            w.cl_start[cid] = (uint32_t)ex; w.cl_n[cid] = (uint32_t)v; w.cl_ikey[cid] = (uint32_t)(ic >> 32); w.toff[h] = (uint32_t)ex;
        }
        carry += rl64(x, 63);
    }
}
// members[] (CSR) from (leader, rank).  One block per scan block: where its leaders' runs start in members[] (cluster start + the run's
// first rank) goes through LDS -- a read's leader is a leader of its own block --, and the claiming leaders wipe their buckets on the way: the
// table is all-zero again when the step ends.
__global__ __launch_bounds__(SB_T) void k_scatter(int64_t n, Work w) {
    __shared__ uint32_t s_dst[SB_READS];
    const int64_t blk = blockIdx.x;
    const uint32_t nl = w.bhdr[blk].n_lead;
    uint32_t v[SB_U];
#pragma unroll
    for (int u = 0; u < SB_U; u++) { const int64_t i = (blk * SB_U + u) * SB_T + threadIdx.x; v[u] = i < n ? w.slot[i] : NONE32; }
    for (uint32_t kq = threadIdx.x; kq < nl; kq += SB_T) {
        const LeadOut o = w.lout[(size_t)blk * SB_READS + kq];
        s_dst[kq] = w.toff[o.h] + (o.rb & ~LO_OWNER);
        if (o.rb & LO_OWNER) *reinterpret_cast<uint4 *>(&w.tab[o.h]) = make_uint4(0, 0, 0, 0);    // (in k_num_reduce, next to the bucket's last read, the same stores cost twice as much: +33 us there, -17 here)
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SB_U; u++) {
        const int64_t i = (blk * SB_U + u) * SB_T + threadIdx.x;
        if (v[u] != NONE32) w.members[s_dst[v[u] & 0xFFFFu] + (v[u] >> 16)] = (uint32_t)i;
    }
}

// ===================================================================================================== read descriptors (not formation)
#ifndef DESC_WPE
#define DESC_WPE 8                       // (64 VGPRs and 44 bytes of scratch per lane; at seven waves -- 72 VGPRs, no scratch -- the kernel is 14 us SLOWER: profiles/r05_y_ab_round_start_vs_head.log)
#endif
__global__ __launch_bounds__(256, DESC_WPE) void k_describe(DevBatch b, DevParams p, Work w, int tiles_per_block) {
    __shared__ long long s_stat[WAVES_PER_BLOCK][6];
    long long st[6] = {0, 0, 0, 0, 0, 0};       // reads, bases, reads_unmapped, bases_unmapped, base_mismatches, reads_with_mismatches
    int lqmin = 0x7FFFFFFF, lqmax = -1;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int cb = 0; cb < tiles_per_block; cb++) {
        const int64_t i = ((int64_t)blockIdx.x * tiles_per_block + cb) * 256 + threadIdx.x;
        if (i < b.n) {
            union { gce_core c; uint4 q[2]; } t; const uint4 *src = reinterpret_cast<const uint4 *>(b.core + i); t.q[0] = src[0]; t.q[1] = src[1];
            const gce_core k = t.c;
            const uint8_t c = d_classify(k);
            bool mapped = k.tid >= 0;                                          // Stats::addRead, stats.cpp:101-121
            const int nmv = b.nm[i]; const bool has_nm = b.nm_type[i] != 0;
            int mism = (mapped && has_nm) ? nmv : 0;
            w.nmx[i] = ((uint32_t)nmv << 1) | (has_nm ? 1u : 0u);              // for the Stats of the emitted records (k_out_meta)
            st[0] += 1; st[1] += k.l_qseq; st[4] += mism;
            if (!mapped) { st[2] += 1; st[3] += k.l_qseq; }
            if (mism > 0) st[5] += 1;
            if (c != CLS_DROP) { lqmin = min(lqmin, k.l_qseq); lqmax = max(lqmax, k.l_qseq); }
            if (c == CLS_CLUSTERED) {
                const uint32_t *cg = b.cigar + b.cigar_off[i];
                const uint32_t c0w = k.n_cigar ? cg[0] : 0;
                int mo_ = 0, ml_ = 0;
                if (k.n_cigar == 1) ml_ = cig_op(c0w) == 0 ? cig_len(c0w) : 0;
                else d_first_m(cg, k.n_cigar, mo_, ml_);
                if (k.l_qseq > 65535) raise_error(w.si, GCE_ERR_INVALID, (uint32_t)i);      // the 16-bit fields of the descriptor (and of the overlap patches)
                if (p.ref_win && k.isize != 0 && k.tid < p.n_ref && p.ref_data[k.tid]) {    // per-shard reference windows: what the vote may look up must be staged
                    const int64_t rl = k.n_cigar == 1 ? (int64_t)cig_len(c0w) * consumes_ref(cig_op(c0w)) : (int64_t)d_cigar_rlen(cg, k.n_cigar);
                    // (a read that overhangs the contig's end is never looked up -- Reference::getData returns NULL, reference.cpp:40,60 -- and the
                    //  staged window cannot reach past the contig: only the bases inside the contig must be there)
                    const int64_t end_in = min((int64_t)k.pos + rl, p.ref_len[k.tid]);
                    if ((int64_t)k.pos < p.ref_win[2 * k.tid] || end_in > p.ref_win[2 * k.tid + 1]) raise_error(w.si, GCE_ERR_REF_WINDOW, (uint32_t)i);
                }
                store_desc(w.rdesc, (uint64_t)i, b.seq_off[i], b.qual_off[i], c0w, k.pos, k.isize != 0, k.l_qseq, mo_, ml_, k.n_cigar, k.tid, k.n_cigar > 1 ? cg[k.n_cigar - 1] : c0w);
                // Pair::setLeft/setRight -> BamUtil::getUMI, bamutil.cpp:23-38
                const char *src2; uint8_t hm = 0;
                if (b.mi && b.mi_off[i] != 0xFFFFFFFFFFFFFFFFull) { src2 = b.mi + b.mi_off[i]; hm = 1; }
                else src2 = b.qname + b.qname_off[i];
                int s0, l0;
                if (!d_umi_slice(src2, p, s0, l0, hm ? -1 : (int)k.l_qname - 1)) { raise_error(w.si, GCE_ERR_UMI_PARSE, (uint32_t)i); s0 = 0; l0 = 0; }
                if (l0 > UINFO_MAXLEN) { raise_error(w.si, GCE_ERR_UMI_PARSE, (uint32_t)i); l0 = 0; }
                w.uinfo[i] = uinfo_pack((hm ? b.mi_off[i] : b.qname_off[i]) + (uint64_t)s0, hm != 0, k.l_qname, l0);
            }
        }
    }
    for (int k = 0; k < 6; k++) { long long v = wave_sum64(st[k]); if (lane == 0) s_stat[wv][k] = v; }
    lqmin = wave_min(lqmin); lqmax = wave_max(lqmax);
    if (lane == 0) {                                                           // (guarded by a plain look: after the first blocks nothing changes any more)
        if (lqmin < *(volatile int *)&w.si->lq_min) atomicMin(&w.si->lq_min, lqmin);
        if (lqmax > *(volatile int *)&w.si->lq_max) atomicMax(&w.si->lq_max, lqmax);
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        long long v = s_stat[0][threadIdx.x] + s_stat[1][threadIdx.x] + s_stat[2][threadIdx.x] + s_stat[3][threadIdx.x];
        if (v) atomicAdd((unsigned long long *)&w.si->pre_slot[blockIdx.x & (GCE_PRE_SLOTS - 1)][threadIdx.x], (unsigned long long)v);
    }
}
