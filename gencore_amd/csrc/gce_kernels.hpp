// gce_kernels.hpp — HIP kernels of the MI355X consensus engine (gfx950, wave64).
//
// Pipeline (one gce_process call, everything resident in HBM; engine.hip launches them in this order):
//   k_cluster        THE CLUSTERING SCAN (gce_cluster.hpp): class, key (tid,left,right), block-level leaders -> (leader, rank) per read
//   k_blk_scan, k_events   ticks of the scan blocks; the reads on which the reference's periodic flush fires (gencore.cpp:319-322)
//   k_leaders        one lane per leader: instance from the flush events, bucket table -> cluster of every leader run
//   k_num_* k_scatter  cluster list (one claiming leader per cluster) + CSR fill: members[] per cluster
//   k_describe       per read: the 32-byte ReadDesc, UMI slice, pre-Stats
//   k_pairing_sub<16|32> (gce_pair2.hpp), k_pairing_fast, k_pairing_deep (gce_deep.hpp), k_pairing_slow
//                    per cluster: qname order, mate pairing (cluster.cpp:260-273), greedy UMI grouping (cluster.cpp:55-100)
//   k_group_fill, k_vote_batches   compact (cluster, group) list, batches of groups
//   k_vote (gce_vote.hpp)   per batch of groups: Pair::computeScore (pair.cpp:88-172) + template pick (group.cpp:136-318) + column
//                    vote (group.cpp:320-579), both sides; groups outside its scope go on through
//   k_score2, k_consensus_fast, k_deep_prepare + k_vote_deep (gce_deep.hpp), k_consensus_slow
//   k_group_tail, k_finish_screen, k_finish   qname reconciliation; duplex merge / filter / FR,RR tags (cluster.cpp:116-188, pair.cpp:43-68)
//   k_out_* (gce_output.hpp)   emitted records in bamComp order as one compact table
//   k_stats          Stats reductions (stats.cpp:101-139)
//   k_depth (gce_depth.hpp)   depth / BED statistics on request
#pragma once
#include "gce_device.hpp"

#define CHUNK 256            // reads per prescan/cluster block
#define WAVES_PER_BLOCK 4

// what Gencore::outputPair hands on for one emitted record (written once, where the record is emitted)
struct __attribute__((aligned(16))) OutRec { uint32_t qname_src, mate; int16_t nm_new, fr, rr; uint16_t pad; };
static_assert(sizeof(OutRec) == 16, "OutRec must stay 16 bytes");

struct Work {
    // per read
    uint64_t *uinfo;                     // per clustered read: where its UMI lies, its length, its name's length -- ONE word (uinfo_pack, below)
    ReadDescP *rdesc;
    uint32_t *spatch;                    // per read: overlap score patch (start | len << 16), GCE_PATCH_CONST, or 0
    uint32_t *slot;                      // per clustered read: leader of its scan block | rank in the leader's run << 16 (gce_cluster.hpp)
    int8_t *score;                       // parallel to qual
    // outputs per read: out_flag for every read (0 = not emitted, 1 = outputPair, 2 = pass-through); orec only where out_flag == 1
    uint8_t *out_flag; OutRec *orec;
    uint32_t *nmx;                       // per read: NM value << 1 | NM present (k_describe)
    uint32_t *out_index;                 // emitted reads, ascending
    // scan blocks (gce_cluster.hpp): leaders of every block, what k_leaders made of them, where their runs start in members[]
    struct LeadRec *lrec; struct LeadOut *lout;
    struct BlkHdr *bhdr; uint32_t *blk_base; int64_t n_sblk;   // per scan block: leaders + clustered reads; clustered reads in front of the block
    // events
    int32_t *ev_tid, *ev_pos; uint32_t *ev_read; int max_events;
    // hash table
    struct TabEntry *tab; uint32_t *toff; uint64_t tsize; double tinv;   // tsize buckets (any size, not a power of two); toff is written sparsely
    // clusters
    uint32_t *cl_ikey, *cl_start, *cl_n, *cl_npairs, *cl_ngroups, *cl_gbase, *cl_nresult; uint8_t *cl_hasumi;
    // cluster-local arrays (indexed by cl_start + k)
    uint32_t *members, *sorted, *pl, *pr, *pu, *pg, *gpl, *gpr, *grp_begin, *grp_n;
    uint64_t *k64;                       // generic pairing scratch: 3 words per read (name window / UMI words)
    // groups (compact)
    uint32_t *gl_cluster, *g_begin, *g_np;   // per compact group: owning cluster, first pair slot, pair count
    uint64_t *gw; uint32_t *g_wbase, *vb_start;   // k_vote batching: weight of every group, its exclusive prefix, first group of every batch
    uint32_t *slow_list;                 // (group*2 + side) entries deferred to the generic consensus kernel
    uint8_t *pf_flag; uint32_t *pf_list;      // clusters the half-wave pairing kernel hands to the full-wave one (same scheme)
    uint8_t *pq_flag; uint32_t *pq_list;      // ... and the quarter-wave kernel to the half-wave one
    uint32_t *left_list;                 // clusters left for k_pairing_deep<device memory> / the generic pairing kernels (count: si->n_slow_pair2).  A list of its own since round 5:
                                         // it is appended to while the half-wave kernel still reads pq_list (the pairing tiers run side by side on two streams)
    uint8_t *p16_flag; uint32_t *p16_list;    // clusters of <= 16 reads, compacted: the quarter-wave kernel runs on full waves (k_pair_classes)
    void *deep_list;                       // DeepRec[] (gce_deep.hpp)
    uint8_t *gen_flag; uint32_t *gen_list;   // (group*2 + side) entries k_vote hands to the per-side kernels: appended to gen_list by the group's lane (one 64-bit atomic per group
                                          // for this list and k_score2's: si->hand_on); gen_flag: 1 = handed on, 2 = finished by k_vote_deep
    uint32_t *score_list;                 // pair slots of the groups whose sides were handed on (k_score2 scores only those; count: low half of si->hand_on)
    uint32_t *rp_left, *rp_right, *rp_merge, *rp_rmerge; const char **rp_umi; uint16_t *rp_umilen; uint8_t *rp_state; int32_t *rp_supp;
    int32_t *rp_nm;                       // [2 x groups] NM byte patched into the side's template (group.cpp:570), -1 = untouched
    uint32_t *rp_qsl, *rp_qsr;            // per group: copyQName source of the left / right result record
    // generic scan scratch
    uint64_t *scan_part;
    StreamInfo *si;
};

// The UMI of a clustered read (BamUtil::getUMI, bamutil.cpp:23-112: a slice of its name, or its MI:Z tag) and the length of its name in ONE 64-bit word per read:
//   bits 0..39 offset of the UMI in the blob it lies in | bit 40: that blob is the MI blob (else the names) | bits 41..48 core.l_qname | bits 49..63 UMI length
// Rounds 1-4 kept a pointer, a 16-bit length and a byte in three arrays: three scattered 64-byte sectors per read wherever a read's UMI was looked at (the
// pairing kernels, k_group_tail) and one more (the core record) for the name's length.  UMIs beyond 32767 bytes are refused by k_describe.
#define UINFO_MAXLEN 32767
__device__ __forceinline__ uint64_t uinfo_pack(uint64_t off, bool from_mi, int l_qname, int len) {
    return (off & 0xFFFFFFFFFFull) | ((uint64_t)(from_mi ? 1 : 0) << 40) | ((uint64_t)(l_qname & 0xFF) << 41) | ((uint64_t)(len & 0x7FFF) << 49);
}
__device__ __forceinline__ const char *uinfo_ptr(const DevBatch &b, uint64_t v) { return (((v >> 40) & 1ull) ? b.mi : b.qname) + (v & 0xFFFFFFFFFFull); }
__device__ __forceinline__ int uinfo_len(uint64_t v) { return (int)(v >> 49); }
__device__ __forceinline__ bool uinfo_hm(uint64_t v) { return ((v >> 40) & 1ull) != 0; }
__device__ __forceinline__ int uinfo_lqname_pad(uint64_t v) { return ((int)((v >> 41) & 0xFFu) + 3) & ~3; }            // (= d_lqname_pad of the read's core record)
__device__ __forceinline__ const char *d_umi_ptr(const DevBatch &b, const Work &w, uint64_t i) { return uinfo_ptr(b, w.uinfo[i]); }
__device__ __forceinline__ int d_umi_len(const Work &w, uint64_t i) { return uinfo_len(w.uinfo[i]); }
__device__ __forceinline__ bool d_umi_hm(const Work &w, uint64_t i) { return uinfo_hm(w.uinfo[i]); }
__device__ __forceinline__ int d_lqname(const Work &w, uint64_t i) { return (int)((w.uinfo[i] >> 41) & 0xFFu); }            // core.l_qname of a clustered read

enum : uint8_t { RP_PENDING = 0, RP_OUT_SSCS = 1, RP_OUT_DCS = 2, RP_DROPPED = 3, RP_CONSUMED = 4 };
// rp_nm: >= 0 the new NM byte, -1 untouched, NM_DEFER + mismatchInc (k_vote): the delta of a template whose NM tag has not been looked at yet
#define NM_DEFER (-0x40000000)
#define NM_IS_DEFERRED(v) ((v) < -0x20000000)

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int rl32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ long long wave_sum64(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

#include "gce_cluster.hpp"

// ===================================================================================================== pairing + UMI grouping
__device__ __forceinline__ const char *d_qname(const DevBatch &b, uint32_t r) { return b.qname + b.qname_off[r]; }

typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ uint64_t d_bswap64(uint64_t x) { return __builtin_bswap64(x); }
// up to 8*NW bytes of a string as big-endian words, zero padded: integer order == strcmp order
template <int NW>
__device__ __forceinline__ void load_be_words(const char *s, int len, uint64_t (&wd)[NW]) {
#pragma unroll
    for (int k = 0; k < NW; k++) {
        uint64_t v = 0;
        int rem = len - 8 * k;
        if (rem > 0) {
            v = *(const u64_unaligned *)(s + 8 * k);
            if (rem < 8) v &= (1ull << (8 * rem)) - 1ull;
            v = d_bswap64(v);
        }
        wd[k] = v;
    }
}
// generic path: any cluster size / name length, cluster-local global scratch
// Three launches (PHASE 0: name windows, 1: ranks, 2: pairs + UMI grouping + layout): the rank scan is O(n^2 / 64) per cluster and
// a cluster of thousands of reads would otherwise keep one wave busy for milliseconds; phase 1 spreads a cluster's 64-read blocks
// over `ibstep` waves.
template <int PHASE>
__device__ void pairing_generic(const DevBatch &b, const DevParams &p, const Work &w, uint32_t c, int lane, uint32_t ib0 = 0, uint32_t ibstep = 1) {
    const uint32_t start = w.cl_start[c], n = w.cl_n[c];
    uint32_t mode = d_thr_mode(w.cl_ikey[c], w.si, p);
    if (mode == THR_NEVER) {                      // pending after an early finishConsensus: never processed (gencore.cpp:23)
        if (PHASE == 2 && lane == 0) { w.cl_npairs[c] = 0; w.cl_ngroups[c] = 0; w.cl_hasumi[c] = 0; }
        return;
    }
    const int thr = mode == THR_PROPER ? p.proper_thr : p.unproper_thr;
    // ---- (a) order the cluster's reads by (qname, input index): map<string,Pair*> order + arrival order (cluster.cpp:260-273)
    //      Rank = number of reads that sort before mine.  The names of a cluster share a long prefix (instrument, run, flowcell);
    //      16 bytes after the cluster's common prefix, as two big-endian words per read in scratch, decide almost every
    //      comparison in registers; equal windows (mates, near-identical names) fall back to strcmp.
    uint64_t *kw = w.k64 + (size_t)start * 3;
    if (PHASE == 0) {
        const char *n0 = d_qname(b, w.members[start]);
        int cp = 0x7FFFFFFF;
        for (uint32_t i = lane; i < n; i += 64) {
            const char *mq = d_qname(b, w.members[start + i]);
            int l = 0;
            while (n0[l] && n0[l] == mq[l]) l++;
            cp = min(cp, l);
        }
        cp = wave_min(cp);
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t my = w.members[start + i];
            const int nl = (int)b.core[my].l_qname - 1;
            uint64_t k2[2];
            load_be_words<2>(d_qname(b, my) + cp, max(nl - cp, 0), k2);
            kw[2 * i] = k2[0]; kw[2 * i + 1] = k2[1];
        }
        return;
    }
    if (PHASE == 1) {
    for (uint32_t base = 64 * ib0; base < n; base += 64 * ibstep) {
        const uint32_t i = base + lane;
        const bool mine = i < n;
        const uint32_t my = mine ? w.members[start + i] : NONE32;
        const char *mq = mine ? d_qname(b, my) : nullptr;
        const uint64_t m0 = mine ? kw[2 * i] : 0ull, m1 = mine ? kw[2 * i + 1] : 0ull;
        uint32_t rk = 0, ntie = 0, t1 = 0, t2 = 0;                           // equal windows (the mate, as a rule) are settled after the scan,
        for (uint32_t jb = 0; jb < n; jb += 64) {                             // every lane on its own: 64 windows per memory round trip, then
            const uint32_t jj = jb + lane;                                    // register broadcasts
            const uint64_t c0 = jj < n ? kw[2 * jj] : 0ull, c1 = jj < n ? kw[2 * jj + 1] : 0ull;
            const int lim = (int)min(64u, n - jb);
            for (int t = 0; t < lim; t++) {
                const uint64_t o0 = rl64(c0, t), o1 = rl64(c1, t);
                const uint32_t j = jb + t;
                rk += (o0 < m0 || (o0 == m0 && o1 < m1));
                if (o0 == m0 && o1 == m1 && j != i) { if (ntie == 0) t1 = j; else if (ntie == 1) t2 = j; ntie++; }
            }
        }
        if (mine) {
            auto before = [&](uint32_t j) { const uint32_t o = w.members[start + j]; const int cmp = d_strcmp(d_qname(b, o), mq); return cmp < 0 || (cmp == 0 && o < my); };
            if (ntie <= 2) { if (ntie >= 1) rk += before(t1); if (ntie == 2) rk += before(t2); }
            else for (uint32_t j = 0; j < n; j++) if (j != i && kw[2 * j] == m0 && kw[2 * j + 1] == m1) rk += before(j);
            w.sorted[start + rk] = my;
        }
    }
    return;
    }
    // ---- (b) pairs: first read of a qname run = mLeft, last of the run (if any other) = mRight (pair.cpp:188-216)
    uint32_t npairs = 0;
    int any_umi = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        uint32_t i = base + lane;
        bool valid = i < n, first = false, last = false;
        uint32_t q = NONE32;
        if (valid) {
            q = w.sorted[start + i];
            const char *mq = d_qname(b, q);
            first = (i == 0) || d_strcmp(d_qname(b, w.sorted[start + i - 1]), mq) != 0;
            last = (i == n - 1) || d_strcmp(d_qname(b, w.sorted[start + i + 1]), mq) != 0;
            if (!first) {       // setRight: `if(!mUMI.empty() && umi!=mUMI) error_exit` (pair.cpp:201-212)
                uint32_t pv = w.sorted[start + i - 1];
                if (d_umi_len(w, pv) != 0 && !d_bytes_equal(d_umi_ptr(b, w, pv), d_umi_len(w, pv), d_umi_ptr(b, w, q), d_umi_len(w, q))) raise_error(w.si, GCE_ERR_UMI_MISMATCH, q);
            }
        }
        unsigned long long fm = __ballot(first);
        uint32_t pidx = npairs + __popcll(fm & ((2ull << lane) - 1ull)) - 1;      // firsts up to and including this lane
        if (valid) {
            if (first) { w.pl[start + pidx] = q; if (last) w.pr[start + pidx] = NONE32; }
            if (last && !first) w.pr[start + pidx] = q;
            if (last) { w.pu[start + pidx] = q; if (d_umi_len(w, q)) any_umi = 1; }
        }
        npairs += __popcll(fm);
    }
    any_umi = __any(any_umi);
    WAVE_SYNC();
    // ---- (c) greedy UMI grouping (cluster.cpp:57-100)
    uint32_t ngroups = 0;
    if (!any_umi) {
        for (uint32_t i = lane; i < npairs; i += 64) w.pg[start + i] = 0;
        ngroups = 1;
    } else {
        uint32_t *pc = w.sorted;                 // reuse: per-pair count of identical UMIs (umiCount)
        // UMIs of <= 24 bytes as three zero-padded words per pair in scratch: equal words <=> equal strings
        bool longu = false;
        for (uint32_t i = lane; i < npairs; i += 64) {
            const uint32_t ui = w.pu[start + i]; const int ul = d_umi_len(w, ui);
            uint64_t u3[3];
            load_be_words<3>(d_umi_ptr(b, w, ui), min(ul, 24), u3);
            kw[3 * i] = u3[0]; kw[3 * i + 1] = u3[1]; kw[3 * i + 2] = u3[2];
            if (ul > 24) longu = true;
        }
        longu = __any(longu);
        WAVE_SYNC();
        for (uint32_t i = lane; i < npairs; i += 64) {
            uint32_t ui = w.pu[start + i];
            const char *up = d_umi_ptr(b, w, ui); int ul = d_umi_len(w, ui);
            uint32_t cnt = 0;
            if (!longu) {
                const uint64_t a0 = kw[3 * i], a1 = kw[3 * i + 1], a2 = kw[3 * i + 2];
                for (uint32_t j = 0; j + 3 < npairs; j += 4) {                    // four independent loads in flight per step
                    const uint64_t x0 = kw[3 * j], x1 = kw[3 * j + 1], x2 = kw[3 * j + 2], y0 = kw[3 * j + 3], y1 = kw[3 * j + 4], y2 = kw[3 * j + 5];
                    const uint64_t z0 = kw[3 * j + 6], z1 = kw[3 * j + 7], z2 = kw[3 * j + 8], v0 = kw[3 * j + 9], v1 = kw[3 * j + 10], v2 = kw[3 * j + 11];
                    cnt += (x0 == a0 && x1 == a1 && x2 == a2) + (y0 == a0 && y1 == a1 && y2 == a2) + (z0 == a0 && z1 == a1 && z2 == a2) + (v0 == a0 && v1 == a1 && v2 == a2);
                }
                for (uint32_t j = npairs & ~3u; j < npairs; j++) cnt += (kw[3 * j] == a0 && kw[3 * j + 1] == a1 && kw[3 * j + 2] == a2);
            } else
            for (uint32_t j = 0; j < npairs; j++) { uint32_t uj = w.pu[start + j]; cnt += d_bytes_equal(up, ul, d_umi_ptr(b, w, uj), d_umi_len(w, uj)); }
            pc[start + i] = cnt;
            w.pg[start + i] = NONE32;
        }
        WAVE_SYNC();
        uint32_t remaining = npairs;
        while (remaining > 0) {
            // top UMI: highest count, lexicographically first on ties (map order + strict `>`, cluster.cpp:68-76)
            uint32_t best = NONE32, bcnt = 0; const char *bu = nullptr; int bl = 0;
            for (uint32_t i = lane; i < npairs; i += 64) {
                if (w.pg[start + i] != NONE32) continue;
                uint32_t ui = w.pu[start + i], cnt = pc[start + i];
                const char *up = d_umi_ptr(b, w, ui); int ul = d_umi_len(w, ui);
                bool better = best == NONE32 || cnt > bcnt || (cnt == bcnt && d_slice_cmp(up, ul, bu, bl) < 0);
                if (better) { best = i; bcnt = cnt; bu = up; bl = ul; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                uint32_t ob = __shfl_xor(best, o), oc = __shfl_xor(bcnt, o);
                if (ob == NONE32) continue;
                uint32_t oui = w.pu[start + ob];
                const char *ou = d_umi_ptr(b, w, oui); int ol = d_umi_len(w, oui);
                bool better;
                if (best == NONE32) better = true;
                else if (oc != bcnt) better = oc > bcnt;
                else { int sc = d_slice_cmp(ou, ol, bu, bl); better = sc < 0 || (sc == 0 && ob < best); }
                if (better) { best = ob; bcnt = oc; bu = ou; bl = ol; }
            }
            uint32_t absorbed = 0;
            for (uint32_t base = 0; base < npairs; base += 64) {
                uint32_t i = base + lane;
                bool take = false;
                if (i < npairs && w.pg[start + i] == NONE32) {
                    uint32_t ui = w.pu[start + i];
                    take = d_umi_diff(d_umi_ptr(b, w, ui), d_umi_len(w, ui), bu, bl) <= thr;
                    if (take) w.pg[start + i] = ngroups;
                }
                absorbed += __popcll(__ballot(take));
            }
            remaining -= absorbed;
            ngroups++;
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
    // ---- (d) lay the pairs out group by group, qname order kept inside a group (Group::addPair, group.cpp:17-22)
    uint32_t gbase = 0;
    for (uint32_t g = 0; g < ngroups; g++) {
        uint32_t run = 0;
        for (uint32_t base = 0; base < npairs; base += 64) {
            uint32_t i = base + lane;
            bool in = i < npairs && w.pg[start + i] == g;
            unsigned long long m = __ballot(in);
            if (in) { uint32_t d = start + gbase + run + lanes_below(m); w.gpl[d] = w.pl[start + i]; w.gpr[d] = w.pr[start + i]; }
            run += __popcll(m);
        }
        if (lane == 0) { w.grp_begin[start + g] = start + gbase; w.grp_n[start + g] = run; }
        gbase += run;
    }
    if (lane == 0) { const bool cross = d_key(b.core[w.members[start]], p).right < 0; w.cl_npairs[c] = npairs; w.cl_ngroups[c] = ngroups; w.cl_hasumi[c] = (uint8_t)((any_umi ? 1 : 0) | (cross ? 2 : 0)); }
}


template <int PHASE>
__global__ __launch_bounds__(256) void k_pairing_slow(DevBatch b, DevParams p, Work w) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t n_slow = w.si->n_slow_pair2;                                      // what k_pairing_deep (gce_deep.hpp) left over
    for (uint32_t idx = blockIdx.x * WAVES_PER_BLOCK + wv; idx < n_slow; idx += gridDim.x * WAVES_PER_BLOCK) {
        const uint32_t c = w.left_list[idx];
        if (c != NONE32) pairing_generic<PHASE>(b, p, w, c, lane, blockIdx.y, gridDim.y);      // (NONE32: taken by the device-memory instantiation of k_pairing_deep)
        WAVE_SYNC();
    }
}

__device__ __forceinline__ int popc_nonzero_bytes(uint64_t x) {
    x |= x >> 4; x |= x >> 2; x |= x >> 1;
    return __popcll(x & 0x0101010101010101ull);
}

// One wave per cluster, <= 64 reads, names <= 64 bytes, UMIs <= 24 bytes: everything in registers.
__device__ void pairing_fast_cluster(const DevBatch &b, const DevParams &p, const Work &w, uint32_t c, int lane) {
    const uint32_t start = w.cl_start[c], n = w.cl_n[c];
    uint32_t mode = d_thr_mode(w.cl_ikey[c], w.si, p);
    if (mode == THR_NEVER) { if (lane == 0) { w.cl_npairs[c] = 0; w.cl_ngroups[c] = 0; w.cl_hasumi[c] = 0; } return; }
    const int thr = mode == THR_PROPER ? p.proper_thr : p.unproper_thr;
    bool defer = n > 64;
    uint32_t my = NONE32; int nl = 0; const char *nm = nullptr; int ul = 0; uint64_t ui_ = 0;
    if (!defer && lane < (int)n) {
        my = w.members[start + lane];
        ui_ = w.uinfo[my];
        nl = (int)((ui_ >> 41) & 0xFFu) - 1;
        nm = d_qname(b, my);
        ul = uinfo_len(ui_);
    }
    if (!defer && __any(nl > 64 || ul > 24)) defer = true;
    if (defer) { if (lane == 0) w.slow_list[atomicAdd(&w.si->n_slow_pair, 1u)] = c; return; }
    const bool act = lane < (int)n;
    uint64_t nw[8];
    load_be_words<8>(nm, act ? nl : 0, nw);
    const int nwords = (wave_max(nl) + 7) >> 3;
    // ---- same-name detection through a 32-bit add-rotate-xor hash of the name words.  It is only a filter: every match is verified
    //      word by word below, and a false match sends the cluster to the generic kernel.  EQ = the other reads of my name (one
    //      ballot per read gives a whole name class at once), LOW = those of them that arrived earlier (smaller input index).
    uint32_t h32 = 0x9E3779B9u;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < nwords) {
        const uint32_t lo = (uint32_t)nw[k], hi = (uint32_t)(nw[k] >> 32);
        h32 = (__builtin_rotateleft32(h32, 5) ^ lo) + hi;
        h32 = __builtin_rotateleft32(h32, 11) ^ (hi + 0x7F4A7C15u);
    }
    h32 ^= h32 >> 15;
    const unsigned long long ACT = __ballot(act);
    unsigned long long EQ = 0, LOW = 0;
    for (unsigned long long todo = ACT; todo;) {             // one ballot per DISTINCT hash: a whole name class at a time
        const int j = __ffsll((long long)todo) - 1;
        const uint32_t oh = (uint32_t)rl32((int)h32, j);
        const unsigned long long cls = __ballot(h32 == oh) & ACT;
        if (h32 == oh) EQ = cls;
        todo &= ~cls;
    }
    EQ &= ~(1ull << lane);
    {   // exact verification of every hash match (all lanes run the shuffles)
        bool bad = false;
        const int rounds = wave_max(act ? __popcll(EQ) : 0);
        unsigned long long rest = act ? EQ : 0ull;
        for (int r = 0; r < rounds; r++) {
            const bool has = rest != 0;
            const int src = has ? __ffsll((long long)rest) - 1 : lane;
            rest &= rest - 1;
            const uint32_t oj = (uint32_t)__shfl((int)my, src);
            if (has && oj < my) LOW |= 1ull << src;
#pragma unroll
            for (int k = 0; k < 8; k++) if (k < nwords) { const uint64_t o = (uint64_t)__shfl((long long)nw[k], src); if (o != nw[k]) bad = true; }
        }
        if (__any(bad)) { if (lane == 0) w.slow_list[atomicAdd(&w.si->n_slow_pair, 1u)] = c; return; }
    }
    // ---- pairs (cluster.cpp:260-273, pair.cpp:188-216): first read of a name = mLeft, last one = mRight
    const bool first = act && !(EQ & LOW), last = act && !(EQ & ~LOW);
    const unsigned long long FIRST = __ballot(first);
    const uint32_t npairs = __popcll(FIRST);
    // ---- map<string,Pair*> order: compares only against the first read of every OTHER name, and on ONE word: the 8 name bytes behind
    //      the cluster's common prefix decide nearly every comparison (the names share instrument / run / lane / tile); names whose
    //      order words tie take the full compare
    unsigned long long LT = 0;
    {
        int cpb = 64;                                       // bytes my name shares with the cluster's first read
#pragma unroll
        for (int k = 7; k >= 0; k--) if (k < nwords) { const uint64_t x = rl64(nw[k], 0) ^ nw[k]; if (x) cpb = 8 * k + (__clzll((long long)x) >> 3); }
        const int cp = min(wave_min(act ? cpb : 64), 56);
        uint64_t okey;
        {
            const int wi = cp >> 3, sh = 8 * (cp & 7);
            uint64_t a = 0, c2 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { if (k == wi) a = nw[k]; if (k == wi + 1) c2 = nw[k]; }
            okey = sh ? ((a << sh) | (c2 >> (64 - sh))) : a;
        }
        unsigned long long TIE = 0;
        for (unsigned long long fm = FIRST; fm; fm &= fm - 1) {
            const int j = __ffsll((long long)fm) - 1;
            const uint64_t o = rl64(okey, j);
            if (o < okey) LT |= 1ull << j;
            if (o == okey && !((EQ >> j) & 1ull) && j != lane) TIE |= 1ull << j;
        }
        for (unsigned long long tm = __ballot(act && TIE != 0) ? FIRST : 0ull; tm; tm &= tm - 1) {      // (rare) ties: whole names, for the lanes that have one with j
            const int j = __ffsll((long long)tm) - 1;
            int cmp = 0;                                    // sign of name_j - name_mine
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k < nwords) {
                    uint64_t o = rl64(nw[k], j);
                    if (cmp == 0) cmp = o < nw[k] ? -1 : (o > nw[k] ? 1 : 0);
                }
            }
            if (((TIE >> j) & 1ull) && cmp < 0) LT |= 1ull << j;
        }
    }
    const uint32_t pidx = __popcll(LT);                     // distinct names before mine
    // every read's UMI as big-endian words in registers (<= 24 bytes, checked above): no byte loops over global memory below
    uint64_t ruw[3];
    load_be_words<3>(act ? uinfo_ptr(b, ui_) : nullptr, act ? ul : 0, ruw);
    {   // setRight (pair.cpp:201-212): the UMI must equal the pair's current UMI if that is non-empty.  The pair's current
        // read is the predecessor in arrival order = the largest read index among the same-name reads before mine.
        unsigned long long prev = act ? (EQ & LOW) : 0ull;
        const int rounds = wave_max(__popcll(prev));
        int plane = -1; uint32_t pv = 0;
        for (int r = 0; r < rounds; r++) {                  // (all lanes run the shuffles)
            const bool has = prev != 0;
            const int src = has ? __ffsll((long long)prev) - 1 : lane;
            prev &= prev - 1;
            const uint32_t o = (uint32_t)__shfl((int)my, src);
            if (has && (plane < 0 || o > pv)) { pv = o; plane = src; }
        }
        if (rounds > 0) {
            const int src = plane < 0 ? lane : plane;
            const uint64_t q0 = (uint64_t)__shfl((long long)ruw[0], src), q1 = (uint64_t)__shfl((long long)ruw[1], src), q2 = (uint64_t)__shfl((long long)ruw[2], src);
            const int qul = __shfl(ul, src);
            if (plane >= 0 && qul != 0 && !(qul == ul && q0 == ruw[0] && q1 == ruw[1] && q2 == ruw[2])) raise_error(w.si, GCE_ERR_UMI_MISMATCH, my);
        }
    }
    const int any_umi = __any(act && last && ul > 0);
    // ---- lanes now stand for pairs (qname order): every read pushes its fields to the lane of its pair (ds_permute).
    //      Lanes that have nothing to send aim at lane 63, which is a pair lane only when all 64 reads are mate-less
    //      singletons -- and then every lane sends.  Unwritten destination lanes read 0.
    const bool pact = lane < (int)npairs;
    uint32_t L = NONE32, R = NONE32, g_of = 0, ngroups = 1;
    {
        const int to_first = (first ? (int)pidx : 63) << 2, to_right = ((last && !first) ? (int)pidx : 63) << 2;
        L = (uint32_t)__builtin_amdgcn_ds_permute(to_first, first ? (int)(my + 1u) : 0) - 1u;
        R = (uint32_t)__builtin_amdgcn_ds_permute(to_right, (last && !first) ? (int)(my + 1u) : 0) - 1u;
        if (!pact) { L = NONE32; R = NONE32; }
    }
    if (any_umi) {                                           // greedy UMI grouping (cluster.cpp:57-100)
        uint64_t uw[3]; int ulen;
        {
            const int to_last = (last ? (int)pidx : 63) << 2;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute(to_last, last ? (int)(uint32_t)ruw[k] : 0);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_permute(to_last, last ? (int)(uint32_t)(ruw[k] >> 32) : 0);
                uw[k] = pact ? (((uint64_t)hi << 32) | lo) : 0ull;
            }
            ulen = __builtin_amdgcn_ds_permute(to_last, last ? ul : 0);
            if (!pact) ulen = 0;
        }
        int cnt = 0, urank = 0;                              // umiCount[umi], and the rank of my UMI in std::string order
        {   // one round per DISTINCT UMI: its pairs learn their count, the pairs with a larger UMI add it to their rank
            const bool one_word = wave_max(ulen) <= 8;       // every UMI fits the first word (the usual 6-8 bp barcode)
            for (unsigned long long todo = __ballot(pact); todo;) {
                const int q = __ffsll((long long)todo) - 1;
                const uint64_t a0 = rl64(uw[0], q); const int al = rl32(ulen, q);
                bool eq, lt;
                if (one_word) { eq = a0 == uw[0] && al == ulen; lt = a0 != uw[0] ? a0 < uw[0] : al < ulen; }
                else {
                    const uint64_t a1 = rl64(uw[1], q), a2 = rl64(uw[2], q);
                    eq = a0 == uw[0] && a1 == uw[1] && a2 == uw[2] && al == ulen;
                    lt = a0 != uw[0] ? a0 < uw[0] : (a1 != uw[1] ? a1 < uw[1] : (a2 != uw[2] ? a2 < uw[2] : al < ulen));
                }
                const unsigned long long cls = __ballot(pact && eq);
                const int sz = __popcll(cls);
                if (pact && eq) cnt = sz;
                if (pact && lt) urank += sz;
                todo &= ~cls;
            }
        }
        g_of = NONE32; ngroups = 0;
        unsigned long long remaining = __ballot(pact);
        while (remaining) {
            int key = (pact && g_of == NONE32) ? (cnt * 64 + (63 - urank)) : -1;     // highest count, then smallest UMI
            int best = wave_max(key);
            int tl = __ffsll((long long)__ballot(key == best)) - 1;
            uint64_t t0 = rl64(uw[0], tl), t1 = rl64(uw[1], tl), t2 = rl64(uw[2], tl);
            int diff = popc_nonzero_bytes(t0 ^ uw[0]) + popc_nonzero_bytes(t1 ^ uw[1]) + popc_nonzero_bytes(t2 ^ uw[2]);   // Cluster::umiDiff
            bool take = pact && g_of == NONE32 && diff <= thr;
            if (take) g_of = ngroups;
            remaining &= ~__ballot(take);
            ngroups++;
        }
    }
    // ---- lay the pairs out group by group (qname order inside a group)
    uint32_t gbase = 0;
    for (uint32_t g = 0; g < ngroups; g++) {
        unsigned long long m = __ballot(pact && g_of == g);
        if (pact && g_of == g) { uint32_t d = start + gbase + lanes_below(m); w.gpl[d] = L; w.gpr[d] = R; }
        uint32_t run = __popcll(m);
        if (lane == 0) { w.grp_begin[start + g] = start + gbase; w.grp_n[start + g] = run; }
        gbase += run;
    }
    if (lane == 0) { const bool cross = d_key(b.core[w.members[start]], p).right < 0; w.cl_npairs[c] = npairs; w.cl_ngroups[c] = ngroups; w.cl_hasumi[c] = (uint8_t)((any_umi ? 1 : 0) | (cross ? 2 : 0)); }
}
// one wave per cluster: every cluster (list == nullptr) or the clusters k_pairing_half (gce_pair2.hpp) flagged, compacted into pf_list
__global__ __launch_bounds__(256) void k_pairing_fast(DevBatch b, DevParams p, Work w, uint32_t n_clusters, const uint32_t *list) {
    const int lane = lane_id();
    const uint32_t total = list ? (uint32_t)w.si->n_pf_items : n_clusters, stride = gridDim.x * WAVES_PER_BLOCK;
    for (uint32_t idx = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6); idx < total; idx += stride) {
        pairing_fast_cluster(b, p, w, list ? list[idx] : idx, lane);
        WAVE_SYNC();
    }
}

// exclusive scan helper over a uint32 array (small-ish n): element = v[i]; reuses the table-scan kernels via tab_elem's low word
__global__ void k_group_fill(Work w, uint32_t n_clusters, int skip_thr, uint32_t deep_weight, uint32_t min_weight) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clusters) return;
    uint32_t g0 = w.cl_gbase[c], ng = w.cl_ngroups[c];
    const uint32_t cs = w.cl_start[c];
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t np = w.grp_n[cs + g];
        w.gl_cluster[g0 + g] = c; w.g_begin[g0 + g] = w.grp_begin[cs + g]; w.g_np[g0 + g] = np;
        // k_vote batches (gce_vote.hpp): a group weighs its pairs (at least VB_MINW: <= 16 groups per batch); a deep group is handed on by a batch of its own
        // (only a group of more than 32 pairs: P0's slot-flag loop is serial in its pairs.  A group that is "deep" by a small --skip_low_complexity
        //  threshold alone is handed on from a shared batch: with a batch of its own for every 2-pair group the batch count left the bound vb_start is sized by)
        (void)skip_thr;
        w.gw[g0 + g] = np > 32u ? (uint64_t)deep_weight : (uint64_t)(np < min_weight ? min_weight : np);
    }
}
// scan of cl_ngroups -> cl_gbase : same 3-phase scheme on the plain counts
__global__ __launch_bounds__(256) void k_u32_apply(const uint32_t *in, uint32_t *out, uint64_t n, const uint64_t *part) {
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = part[blockIdx.x];
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        uint64_t h = base + k * 256 + threadIdx.x;
        uint64_t v = tab_elem(in, h, n), x = v;
        for (int o = 1; o < 64; o <<= 1) { uint64_t t = (uint64_t)__shfl_up((long long)x, o); if (lane >= o) x += t; }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        uint64_t woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        uint64_t carry = s_carry;
        if (h < n) out[h] = (uint32_t)(carry + woff + x - v);
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + woff + x;
        __syncthreads();
    }
}

// ===================================================================================================== scoring
// Pair::computeScore (pair.cpp:88-172).  gpl/gpr are pre-filled with NONE32, so a slot is a pair iff gpl != NONE32.
// Pairs of groups that never reach a vote get scores too; nothing reads them and no qual is touched for a mate-less
// pair, so the result is identical to the reference's lazy evaluation.
// Only the mate-overlap region needs work: outside it a score is qual2score(qual) of an untouched qual, which the vote kernels
// derive on the fly (d_q2s4_biased / d_score_at).  Per pair the kernel writes the two reads' patch descriptors and, for
// overlapping pairs, the overlap scores (match: qual2score((lq+rq)/2)+4; mismatch: quals rewritten to max(0, own - mate),
// the stronger side gets qual2score(|diff|)-3, the other 0).
// ONE LANE PER PAIR for the dependent part (slot -> reads -> descriptors -> overlap
// window and patch descriptors: 64 pairs share each round trip instead of 8), then the wave's overlap work is cut into units of
// 8 bases and dealt evenly to the lanes (half of the pairs have no overlap, a few have a long one): unit -> pair through a small
// LDS map, the pair's fields through ds_bpermute.  A unit costs one 8-byte load per array and side and one 8-byte score store
// per side; only complete units are stored as words (the bytes behind the last overlap base belong to other patches).
#define SC2_MAXU 1280        // >= 64 pairs x ceil(150 / 8); longer overlaps take more rounds of the unit map
__global__ __launch_bounds__(256) void k_score2(DevBatch b, DevParams p, Work w) {
    __shared__ uint8_t s_map[WAVES_PER_BLOCK][SC2_MAXU];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t n_list = (uint32_t)w.si->hand_on;
    // a wave takes 64 entries of the list at a time (round 5: a wave per 64 SLOTS of the stream, 78 k blocks looking at flags that are nearly all zero, was 40 of the kernel's 57 us at cfg3)
    for (uint32_t base = (blockIdx.x * WAVES_PER_BLOCK + wv) * 64u; base < n_list; base += gridDim.x * WAVES_PER_BLOCK * 64u) {
    const uint32_t li = base + (uint32_t)lane;
    uint32_t L = NONE32, R = NONE32;
    if (li < n_list) { const uint32_t slot = w.score_list[li]; L = w.gpl[slot]; if (L != NONE32) R = w.gpr[slot]; }
    int lstart = 0, rstart = 0, cmp = 0;
    uint64_t lso = 0, rso = 0, lqo = 0, rqo = 0;
    if (L != NONE32) {
        if (R == NONE32) w.spatch[L] = GCE_PATCH_CONST;                                          // pair.cpp:89-105
        else {
            const ReadDesc lk = load_desc(w.rdesc, L), rk = load_desc(w.rdesc, R);
            if (!(lk.ml > 0 && rk.ml > 0)) { w.spatch[L] = GCE_PATCH_CONST; w.spatch[R] = GCE_PATCH_CONST; }
            else {
                const int dis = rk.pos - lk.pos;                                                 // pair.cpp:108-120
                if (dis >= 0) { lstart = lk.mo + dis; rstart = rk.mo; cmp = min(lk.ml - dis, rk.ml); }
                else { lstart = lk.mo; rstart = rk.mo - dis; cmp = min(lk.ml, rk.ml + dis); }
                if (cmp > 0 && (lk.lq > 65535 || rk.lq > 65535)) { raise_error(w.si, GCE_ERR_INVALID, L); cmp = 0; }
                if (cmp > 0) {
                    w.spatch[L] = (uint32_t)lstart | ((uint32_t)cmp << 16); w.spatch[R] = (uint32_t)rstart | ((uint32_t)cmp << 16);
                    lso = lk.so; rso = rk.so; lqo = lk.qo; rqo = rk.qo;
                } else { cmp = 0; w.spatch[L] = 0u; w.spatch[R] = 0u; }                          // no overlap: both reads are pure qual2score (written, not assumed: nobody clears spatch)
            }
        }
    }
    // ---- units of 8 overlap bases, exclusive prefix over the wave's pairs
    const int nu = (cmp + 7) >> 3;
    int pre = nu;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, o); if (lane >= o) pre += t; }
    const int total = __shfl(pre, 63);
    pre -= nu;
    for (int ubase = 0; ubase < total; ubase += SC2_MAXU) {                                      // (one pass unless the overlaps are huge)
        for (int k = 0; k < nu; k++) { const int u = pre + k - ubase; if (u >= 0 && u < SC2_MAXU) s_map[wv][u] = (uint8_t)lane; }
        WAVE_SYNC();
        const int lim = min(SC2_MAXU, total - ubase);
        for (int r0 = 0; r0 < lim; r0 += 64) {
            const int u = r0 + lane;
            const bool live = u < lim;
            const int pl = live ? s_map[wv][u] : 0;                                              // (all lanes run the shuffles)
            const int ppre = __shfl(pre, pl), pcmp = __shfl(cmp, pl), pls = __shfl(lstart, pl), prs = __shfl(rstart, pl);
            const uint64_t plso = (uint64_t)__shfl((long long)lso, pl), prso = (uint64_t)__shfl((long long)rso, pl);
            const uint64_t plqo = (uint64_t)__shfl((long long)lqo, pl), prqo = (uint64_t)__shfl((long long)rqo, pl);
            if (live) {
                const int i0 = 8 * (ubase + u - ppre), nv = min(8, pcmp - i0), l = pls + i0, r = prs + i0;
                uint8_t *lq = b.qual + plqo, *rq = b.qual + prqo;
                const uint64_t ql8 = *(const u64_unaligned *)(lq + l), qr8 = *(const u64_unaligned *)(rq + r);
                const uint64_t sl8 = *(const u64_unaligned *)(b.seq + plso + (l >> 1)), sr8 = *(const u64_unaligned *)(b.seq + prso + (r >> 1));
                uint64_t outl = 0, outr = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int ql = (int)(ql8 >> (8 * k)) & 0xFF, qr = (int)(qr8 >> (8 * k)) & 0xFF;
                    const int jl = (l & 1) + k, jr = (r & 1) + k;                             // nibble index in the loaded bytes: byte j/2, high nibble first
                    const int nl = (int)(sl8 >> (8 * (jl >> 1) + ((jl & 1) ? 0 : 4))) & 0xF, nr = (int)(sr8 >> (8 * (jr >> 1) + ((jr & 1) ? 0 : 4))) & 0xF;
                    int scl, scr;
                    if (nl == nr) scl = scr = d_qual2score(p, ((ql + qr) / 2) & 0xFF) + 4 + p.score_bias;        // pair.cpp:148-154
                    else {                                                                     // pair.cpp:155-168: quals rewritten in place
                        if (k < nv) { lq[l + k] = (uint8_t)max(0, ql - qr); rq[r + k] = (uint8_t)max(0, qr - ql); }
                        if (ql >= qr) { scl = d_qual2score(p, ql - qr) - 3 + p.score_bias; scr = p.score_bias; }
                        else { scl = p.score_bias; scr = d_qual2score(p, qr - ql) - 3 + p.score_bias; }
                    }
                    outl |= (uint64_t)(scl & 0xFF) << (8 * k); outr |= (uint64_t)(scr & 0xFF) << (8 * k);
                }
                int8_t *ls = w.score + plqo, *rs = w.score + prqo;
                if (nv == 8) { *(u64_unaligned *)(ls + l) = outl; *(u64_unaligned *)(rs + r) = outr; }
                else for (int k = 0; k < nv; k++) { ls[l + k] = (int8_t)(outl >> (8 * k)); rs[r + k] = (int8_t)(outr >> (8 * k)); }
            }
        }
        WAVE_SYNC();
    }
    }
}

// ===================================================================================================== consensus
struct VoteCtx {
    const DevBatch *b; const DevParams *p; const Work *w;
    uint32_t out, nv, vbase; bool left_mode;
    int len; const uint8_t *ref; int64_t ref_len;
    const uint32_t *ocig; int oncig; int opos;
    uint32_t *tally;      // lane-private LDS: word (bin*3+k)*64
    const uint32_t *voters; const uint32_t *vld;   // scratch: voter reads and their lenDiff
};
struct ColResult { int base, qual, new_base_written, minc, diff; };

// One column of Group::makeConsensus (group.cpp:369-526) for column `col` of the template.
__device__ inline ColResult vote_column(const VoteCtx &v, int col, int out_base, int lane, bool active = true) {
    const DevBatch &b = *v.b; const DevParams &p = *v.p; const Work &w = *v.w;
    uint32_t *t = v.tally + lane;
#pragma unroll
    for (int k = 0; k < 48; k++) t[k * 64] = 0;
    int total = 0;
    // voters in chunks of 64: every lane fetches one voter's offsets / length / patch (the dependent part of the chain, once per
    // chunk), the inner loop broadcasts them from registers, so a voter costs one round trip for its bytes instead of three.
    // (This function is called from wave-uniform code only: all lanes run the broadcasts.)
    for (uint32_t qb = 0; qb < v.nv; qb += 64) {
        const uint32_t qq = qb + (uint32_t)lane;
        uint64_t m_so = 0, m_qo = 0; int m_rl = 0, m_ld = 0; uint32_t m_patch = 0;
        if (qq < v.nv) {
            const uint32_t r = v.voters[v.vbase + qq];
            m_ld = (int)v.vld[v.vbase + qq]; m_rl = b.core[r].l_qseq; m_so = b.seq_off[r]; m_qo = b.qual_off[r]; m_patch = w.spatch[r];
        }
        const int lim = (int)min(64u, v.nv - qb);
        for (int tq_ = 0; tq_ < lim; tq_++) {
            const uint64_t so = rl64(m_so, tq_), qo = rl64(m_qo, tq_);
            const int rl = rl32(m_rl, tq_), ld = rl32(m_ld, tq_); const uint32_t patch = (uint32_t)rl32((int)m_patch, tq_);
            const int rp = v.left_mode ? col : col + ld;
            if (active && rp >= 0 && rp < rl) {                 // out of range is UB in the reference; skipped (same as oracle)
                int base = d_nib(b.seq + so, rp);
                int qu = b.qual[qo + rp];
                int sc = d_score_at(p, w.score + qo, patch, rp, qu);
                uint32_t t0 = t[(base * 3) * 64];
                uint32_t cnt = (t0 & 0xFFFF) + 1, tq = t0 >> 16;
                if ((uint32_t)qu > tq) tq = qu;
                t[(base * 3) * 64] = cnt | (tq << 16);
                t[(base * 3 + 1) * 64] += (uint32_t)sc;
                t[(base * 3 + 2) * 64] += (uint32_t)qu;
                total += sc;
            }
        }
    }
    if (!active) { ColResult none; none.base = out_base; none.qual = 0; none.new_base_written = 0; none.minc = 0; none.diff = 0; return none; }
    // top / second base: lexicographic max of (score, sum of quals), the LATER bin wins ties (group.cpp:394-416, quirk Q6)
    int top = 0, top_s = -0x7FFFFFFF, top_q = 0;
    for (int bb = 0; bb < 16; bb++) {
        int s = (int)t[(bb * 3 + 1) * 64], qs = (int)t[(bb * 3 + 2) * 64];
        if (s > top_s || (s == top_s && qs >= top_q)) { top_s = s; top = bb; top_q = qs; }
    }
    int sec = 0, sec_s = -0x7FFFFFFF, sec_q = (int)t[2 * 64];     // quals[secBase] with secBase initially 0
    for (int bb = 0; bb < 16; bb++) {
        if (bb == top) continue;
        int s = (int)t[(bb * 3 + 1) * 64], qs = (int)t[(bb * 3 + 2) * 64];
        if (s > sec_s || (s == sec_s && qs >= sec_q)) { sec_s = s; sec = bb; sec_q = qs; }
    }
    uint32_t tt = t[(top * 3) * 64];
    int top_num = tt & 0xFFFF, top_qual = tt >> 16;
    int sec_num = t[(sec * 3) * 64] & 0xFFFF;
    sec_q = (int)t[(sec * 3 + 2) * 64];
    ColResult res; res.minc = 0; res.diff = 0; res.new_base_written = 0; res.base = out_base;
    bool need = false;
    if (sec_num == 0) {                                                       // group.cpp:421-428
        if (top_s >= p.base_score_req && top_qual >= p.moderate_q) { res.qual = top_qual; return res; }
        need = true;
    }
    int ref4 = 0;                                                             // group.cpp:430-439
    if (v.ref) {
        int ro = d_ref_offset(v.ocig, v.oncig, col);
        if (ro >= 0 && (int64_t)v.opos + ro < v.ref_len) ref4 = d_ref_nib(v.ref, (int64_t)v.opos + ro);
    }
    if (sec_num == 1) {                                                       // group.cpp:442-457
        if (sec_q <= p.low_q) { if (top_num < 2 && top_qual < p.high_q) need = true; }
        else { if (top_num < 3 || top_qual < p.high_q) need = true; }
    }
    if (sec_num > 1) {                                                        // group.cpp:460-464 (double, quirk Q12)
        if ((double)top_s < p.score_percent_req * (double)total || top_qual < p.moderate_q) need = true;
    }
    if (top_s < p.base_score_req || top_qual <= p.low_q) need = true;         // group.cpp:466-467
    if (need && ref4 != 0) {                                                  // group.cpp:470-501
        int rbq = 0;                                                          // `char refBaseQual`
        for (uint32_t q = 0; q < v.nv; q++) {
            uint32_t r = v.voters[v.vbase + q];
            int ld = (int)v.vld[v.vbase + q];
            int rl = b.core[r].l_qseq;
            int rp = v.left_mode ? col : col + ld;
            if (rp < 0 || rp >= rl) continue;
            int base = d_nib(b.seq + b.seq_off[r], rp);
            int qu = b.qual[b.qual_off[r] + rp];
            if (base == ref4) {
                if (qu > rbq) rbq = (int)(int8_t)qu;
                if (qu >= p.high_q) top = ref4;
            }
        }
        if (top_qual < p.moderate_q) top = ref4;
        if (top == ref4) top_qual = rbq & 0xFF;
    }
    if (out_base != top) {                                                    // group.cpp:503-524
        res.base = top; res.new_base_written = 1; res.diff = 1;
        if (ref4 != 0) { if (out_base == ref4) res.minc = 1; else if (top == ref4) res.minc = -1; }
    }
    res.qual = top_qual;
    return res;
}

// Scratch (cluster-local arrays that are dead after k_pairing): left side uses sorted/pl/pr, right side pg/pu/members
// for containedBy / voters / lenDiff (the two sides of a group may run concurrently on different waves).
struct SidePrep {                       // what Group::consensusMergeBam hands to makeConsensus (group.cpp:268-315)
    uint32_t out, nv; bool left_mode; int len; const uint8_t *ref; int64_t ref_len;
    const uint32_t *voters, *vld;       // cluster-local scratch: voter reads (template first) and their lenDiff, at [begin, begin + nv)
};
// template pick + voter list of one group side (wave-level).  prep.out == NONE32: the side yields no read.
__device__ SidePrep side_prepare(const DevBatch &b, const DevParams &p, const Work &w, uint32_t begin, uint32_t np, bool is_left, int lane) {
    SidePrep sp; sp.out = NONE32; sp.nv = 0; sp.left_mode = is_left; sp.len = 0; sp.ref = nullptr; sp.ref_len = 0; sp.voters = nullptr; sp.vld = nullptr;
    const uint32_t *side = is_left ? w.gpl : w.gpr;
    // ---- leftReadMode (group.cpp:177-194)
    bool left_mode = is_left;
    if (!is_left) {
        int mn = 0x7FFFFFFF, mx = -1;
        for (uint32_t k0 = lane; k0 < np; k0 += 256) {                          // (four batches in flight: index -> position is two dependent round trips)
            uint32_t rdv[4]; int psv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t k = k0 + 64u * u; rdv[u] = k < np ? side[begin + k] : NONE32; }
#pragma unroll
            for (int u = 0; u < 4; u++) psv[u] = b.core[rdv[u] != NONE32 ? rdv[u] : 0u].pos;
#pragma unroll
            for (int u = 0; u < 4; u++) if (rdv[u] != NONE32) { mn = min(mn, psv[u]); mx = max(mx, psv[u]); }
        }
        mn = wave_min(mn); mx = wave_max(mx);
        if (mx < 0 || mn == mx) left_mode = true;
    }
    // ---- containedBy (group.cpp:196-233)
    uint32_t *contained = is_left ? w.sorted : w.pg;
    uint32_t *voters = is_left ? w.pl : w.pu, *vld = is_left ? w.pr : w.members;
    // Deep sides first try CLASSES: reads with the same CIGAR (and, right side, the same right end) are interchangeable for
    // isPartOf, so containedBy(read) = sum over classes of |class| x [read is part of the class] -- O(reads x classes) on
    // registers instead of O(reads^2) walks through memory.  One class per lane (<= 64), CIGARs of <= 4 ops; anything else
    // takes the pairwise loop below.
    bool classed = false, lowc_done = false;
    // a read of the side as the class logic sees it: CIGAR length, right end (right sides), the <= 4 CIGAR words ORIENTED from the compared
    // end (bamutil.cpp:213-218), length, position.  Four reads per lane are fetched at a time (index -> descriptor -> CIGAR words is a
    // dependent chain of two or three round trips; a wave that walked its side 64 reads per trip spent its life waiting for them).
    struct RD { bool has; int n, rr, lq, pos; uint32_t w0, w1, w2, w3, rd; };
    auto fill_rd = [&](RD &r, uint32_t rd, const ReadDesc &d) {
        r.has = rd != NONE32; r.n = 0; r.rr = 0; r.lq = 0; r.pos = 0; r.w0 = r.w1 = r.w2 = r.w3 = 0; r.rd = rd;
        if (!r.has) return;
        r.lq = d.lq; r.pos = d.pos; r.n = d.nc;
        r.rr = is_left ? 0 : d.pos + (d.rlen != RLEN_WALK ? d.rlen : d_cigar_rlen(b.cigar + b.cigar_off[rd], d.nc));
        if (r.n == 1) r.w0 = d.c0;
        else if (r.n >= 2 && r.n <= 4) {
            const uint32_t *cg = b.cigar + b.cigar_off[rd];
            r.w0 = left_mode ? cg[0] : cg[r.n - 1]; r.w1 = left_mode ? cg[1] : cg[r.n - 2];
            if (r.n >= 3) r.w2 = left_mode ? cg[2] : cg[r.n - 3];
            if (r.n >= 4) r.w3 = left_mode ? cg[3] : cg[0];
        }
    };
    auto load_read1 = [&](uint32_t k) {
        RD r; const uint32_t rd = k < np ? side[begin + k] : NONE32;
        const ReadDesc d = load_desc(w.rdesc, rd != NONE32 ? rd : 0u);
        fill_rd(r, rd, d);
        return r;
    };
    auto load_read4 = [&](uint32_t base, RD (&r)[4]) {
        uint32_t rdv[4]; ReadDesc d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = base + 64u * u + lane; rdv[u] = k < np ? side[begin + k] : NONE32; }
#pragma unroll
        for (int u = 0; u < 4; u++) d[u] = load_desc(w.rdesc, rdv[u] != NONE32 ? rdv[u] : 0u);
#pragma unroll
        for (int u = 0; u < 4; u++) fill_rd(r[u], rdv[u], d[u]);
    };
    auto part_of4 = [&](int pn, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, int wn, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
        if (wn < pn) return false;                                      // BamUtil::isPartOf (bamutil.cpp:204-255) on oriented words
        const uint32_t pw[4] = {p0, p1, p2, p3}, ww[4] = {q0, q1, q2, q3};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < pn) {
                if (cig_op(pw[i]) != cig_op(ww[i])) return false;
                const int la = cig_len(pw[i]), lb = cig_len(ww[i]);
                if (la > lb) return false;
                if (la < lb && i != pn - 1) {
                    if (i != pn - 2) return false;
                    if (cig_op(pw[i + 1 < 4 ? i + 1 : 3]) != 5 /*H*/) return false;
                }
            }
        }
        return true;
    };
    uint32_t first_read = NONE32;                  // firstRead of group.cpp:145-160: the side's first present read
    if (np > 64) {
        int c_n = 0, c_cnt = 0, c_rr = 0; uint32_t c_w0 = 0, c_w1 = 0, c_w2 = 0, c_w3 = 0;      // lane c = class c
        int nclass = 0; bool fail = false;
        for (uint32_t base4 = 0; base4 < np && !fail; base4 += 256) {
          RD r4[4];
          load_read4(base4, r4);
#pragma unroll
          for (int u4 = 0; u4 < 4; u4++) {
            const uint32_t base = base4 + 64u * u4;
            if (base >= np || fail) break;
            const bool has = r4[u4].has; const int n_ = r4[u4].n, rr_ = r4[u4].rr; const uint32_t w0 = r4[u4].w0, w1 = r4[u4].w1, w2 = r4[u4].w2, w3 = r4[u4].w3;
            if (__any(has && n_ > 4)) { fail = true; break; }
            const unsigned long long hm = __ballot(has);
            if (first_read == NONE32 && hm) first_read = side[begin + base + (__ffsll((long long)hm) - 1)];
            // against the classes known so far: all reads of the batch at once, one broadcast per CLASS (a read at a time -- 64 rounds of six
            // broadcasts and a ballot per batch -- was most of this kernel's instruction stream: a deep side has a handful of classes)
            bool placed = !has;
            for (int cc = 0; cc < nclass; cc++) {
                const int q_n = rl32(c_n, cc), q_rr = rl32(c_rr, cc);
                const uint32_t q0 = (uint32_t)rl32((int)c_w0, cc), q1 = (uint32_t)rl32((int)c_w1, cc), q2 = (uint32_t)rl32((int)c_w2, cc), q3 = (uint32_t)rl32((int)c_w3, cc);
                const bool eq = !placed && n_ == q_n && rr_ == q_rr && w0 == q0 && w1 == q1 && w2 == q2 && w3 == q3;
                const unsigned long long hit = __ballot(eq);
                if (lane == cc) c_cnt += __popcll(hit);
                placed |= eq;
            }
            // reads of a class not seen before, one at a time (a later one may belong to the class an earlier one opens)
            for (unsigned long long m = __ballot(!placed); m; m &= m - 1) {
                const int t = __ffsll((long long)m) - 1;
                const int r_n = rl32(n_, t), r_rr = rl32(rr_, t);
                const uint32_t r0 = (uint32_t)rl32((int)w0, t), r1 = (uint32_t)rl32((int)w1, t), r2 = (uint32_t)rl32((int)w2, t), r3 = (uint32_t)rl32((int)w3, t);
                const unsigned long long hit = __ballot(lane < nclass && c_n == r_n && c_rr == r_rr && c_w0 == r0 && c_w1 == r1 && c_w2 == r2 && c_w3 == r3);
                if (hit) { if (lane == __ffsll((long long)hit) - 1) c_cnt++; }
                else {
                    if (nclass == 64) { fail = true; break; }
                    if (lane == nclass) { c_n = r_n; c_rr = r_rr; c_w0 = r0; c_w1 = r1; c_w2 = r2; c_w3 = r3; c_cnt = 1; }
                    nclass++;
                }
            }
          }
        }
        // ---- low-complexity skip for very deep groups (group.cpp:142-175), from the classes: distinct CIGAR strings = classes whose
        //      words differ from every earlier class (a right side keys its classes by right end as well)
        if (!fail && (int)np > p.skip_low_complexity_thr) {
            bool dup = false;
            for (int cc = 0; cc < nclass; cc++) {
                const int q_n = rl32(c_n, cc);
                const uint32_t q0 = (uint32_t)rl32((int)c_w0, cc), q1 = (uint32_t)rl32((int)c_w1, cc), q2 = (uint32_t)rl32((int)c_w2, cc), q3 = (uint32_t)rl32((int)c_w3, cc);
                if (lane > cc && lane < nclass && c_n == q_n && c_w0 == q0 && c_w1 == q1 && c_w2 == q2 && c_w3 == q3) dup = true;
            }
            const int distinct = __popcll(__ballot(lane < nclass && !dup));
            lowc_done = true;
            if ((double)distinct > (double)np * 0.1 && first_read != NONE32) {
                int n = b.core[first_read].l_qseq, dn = 0;
                const uint8_t *s = b.seq + b.seq_off[first_read];
                for (int i = lane; i < n - 1; i += 64) dn += d_base_class(d_nib(s, i)) != d_base_class(d_nib(s, i + 1));
                dn = wave_sum(dn);
                if ((double)dn < (double)n * 0.5) return sp;
            }
        }
        if (!fail) {
            for (uint32_t base4 = 0; base4 < np; base4 += 256) {
                RD r4[4];
                load_read4(base4, r4);
                uint32_t cb[4] = {0, 0, 0, 0};
                for (int cc = 0; cc < nclass; cc++) {                               // (one broadcast of a class serves the four reads of the lane)
                    const int q_n = rl32(c_n, cc), q_rr = rl32(c_rr, cc), q_cnt = rl32(c_cnt, cc);
                    const uint32_t q0 = (uint32_t)rl32((int)c_w0, cc), q1 = (uint32_t)rl32((int)c_w1, cc), q2 = (uint32_t)rl32((int)c_w2, cc), q3 = (uint32_t)rl32((int)c_w3, cc);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (r4[u].has && (is_left || r4[u].rr == q_rr) && part_of4(r4[u].n, r4[u].w0, r4[u].w1, r4[u].w2, r4[u].w3, q_n, q0, q1, q2, q3)) cb[u] += (uint32_t)q_cnt;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t k = base4 + 64u * u + lane; if (k < np) { contained[begin + k] = r4[u].has ? cb[u] : 0; vld[begin + k] = (uint32_t)r4[u].lq; } }   // (vld: the read lengths, for the template pick below; the voter list overwrites them later)
            }
            classed = true;
        }
    }
    // ---- low-complexity skip for very deep groups (group.cpp:142-175)
    if (!lowc_done && (int)np > p.skip_low_complexity_thr) {     // (> 64 classes or CIGARs of > 4 ops: pairwise)
        int distinct = 0; uint32_t first = NONE32;
        for (uint32_t base = 0; base < np; base += 64) {
            uint32_t k = base + lane;
            bool uniq = false; uint32_t rd = NONE32;
            if (k < np && (rd = side[begin + k]) != NONE32) {
                uniq = true;
                int nc = b.core[rd].n_cigar; const uint32_t *cg = b.cigar + b.cigar_off[rd];
                for (uint32_t j = 0; j < k && uniq; j++) {
                    uint32_t o = side[begin + j];
                    if (o == NONE32 || b.core[o].n_cigar != nc) continue;
                    const uint32_t *og = b.cigar + b.cigar_off[o];
                    bool same = true;
                    for (int x = 0; x < nc; x++) same &= (og[x] == cg[x]);
                    if (same) uniq = false;
                }
            }
            unsigned long long has = __ballot(rd != NONE32);
            if (first == NONE32 && has) first = side[begin + base + (__ffsll((long long)has) - 1)];
            distinct += __popcll(__ballot(uniq));
        }
        if ((double)distinct > (double)np * 0.1 && first != NONE32) {
            int n = b.core[first].l_qseq, dn = 0;
            const uint8_t *s = b.seq + b.seq_off[first];
            for (int i = lane; i < n - 1; i += 64) dn += d_base_class(d_nib(s, i)) != d_base_class(d_nib(s, i + 1));
            dn = wave_sum(dn);
            if ((double)dn < (double)n * 0.5) return sp;
        }
    }
    if (!classed)
    for (uint32_t base = 0; base < np; base += 64) {
        uint32_t k = base + lane;
        if (k < np) {
            uint32_t part = side[begin + k];
            uint32_t cb = 0;
            if (part != NONE32) {
                cb = 1;
                gce_core pk = b.core[part]; const uint32_t *pc = b.cigar + b.cigar_off[part];
                int prr = is_left ? 0 : pk.pos + d_cigar_rlen(pc, pk.n_cigar);
                for (uint32_t j = 0; j < np; j++) {
                    if (j == k) continue;
                    uint32_t whole = side[begin + j];
                    if (whole == NONE32) continue;
                    gce_core wk = b.core[whole]; const uint32_t *wc = b.cigar + b.cigar_off[whole];
                    if (!is_left && prr != wk.pos + d_cigar_rlen(wc, wk.n_cigar)) continue;
                    if (d_is_part_of(pc, pk.n_cigar, wc, wk.n_cigar, left_mode)) cb++;
                }
            }
            contained[begin + k] = cb;
        }
    }
    WAVE_SYNC();
    if ((int)np > p.skip_low_complexity_thr) {                 // the early `break` at group.cpp:231-232: later entries stay 0
        uint32_t stop = NONE32;
        for (uint32_t base = 0; base < np && stop == NONE32; base += 64) {
            uint32_t k = base + lane;
            unsigned long long m = __ballot(k < np && contained[begin + k] >= np / 2);
            if (m) stop = base + (__ffsll((long long)m) - 1);
        }
        if (stop != NONE32) for (uint32_t k = stop + 1 + lane; k < np; k += 64) contained[begin + k] = 0;
        WAVE_SYNC();
    }
    // ---- template = max containedBy, then shorter read, then first in qname order (group.cpp:235-261)
    uint32_t best = NONE32; int bc = -1, blen = 0;
    if (classed) {                                                          // (lengths next to the counts: two coalesced loads, four batches in flight)
        for (uint32_t k0 = lane; k0 < np; k0 += 256) {
            int cbv[4], lnv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t k = k0 + 64u * u; cbv[u] = k < np ? (int)contained[begin + k] : -1; lnv[u] = k < np ? (int)vld[begin + k] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = k0 + 64u * u;
                if (k < np && (best == NONE32 || cbv[u] > bc || (cbv[u] == bc && lnv[u] < blen))) { best = k; bc = cbv[u]; blen = lnv[u]; }
            }
        }
    } else
    for (uint32_t k = lane; k < np; k += 64) {
        int cb = (int)contained[begin + k];
        uint32_t rd = side[begin + k];
        int ln = rd != NONE32 ? b.core[rd].l_qseq : 0;
        if (best == NONE32 || cb > bc || (cb == bc && ln < blen)) { best = k; bc = cb; blen = ln; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t ob = __shfl_xor(best, o); int oc = __shfl_xor(bc, o), ol = __shfl_xor(blen, o);
        if (ob == NONE32) continue;
        bool better = best == NONE32 || oc > bc || (oc == bc && (ol < blen || (ol == blen && ob < best)));
        if (better) { best = ob; bc = oc; blen = ol; }
    }
    if ((double)bc < (double)np * 0.4 && np != 1) return sp;            // group.cpp:264-266
    uint32_t out = side[begin + best];
    if (out == NONE32) return sp;                                        // group.cpp:283-285
    gce_core ok = b.core[out];
    const uint32_t *ocig = b.cigar + b.cigar_off[out];
    // ---- voters: template + every read the template is part of (group.cpp:287-313); lenDiff (group.cpp:339-348)
    WAVE_SYNC();
    uint32_t nv = 1;
    if (lane == 0) { voters[begin] = out; vld[begin] = 0; }
    if (classed) {
        // the classes' reads have <= 4 CIGAR ops: isPartOf on the oriented words in registers (as containedBy above), length and position
        // from the descriptor -- not a gather of the alignment record and two CIGAR walks through memory per read
        const RD t = load_read1(best);                                          // (every lane: the template)
        for (uint32_t base4 = 0; base4 < np; base4 += 256) {
            RD r4[4];
            load_read4(base4, r4);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = base4 + 64u * u + lane;
                if (base4 + 64u * u >= np) break;
                const RD &r = r4[u];
                const bool take = r.has && j != best && part_of4(t.n, t.w0, t.w1, t.w2, t.w3, r.n, r.w0, r.w1, r.w2, r.w3);
                int ld = 0;
                if (take) {
                    ld = r.lq - ok.l_qseq;
                    if (ld != 0 && r.pos == ok.pos && (left_mode || d_is_part_of(ocig, ok.n_cigar, b.cigar + b.cigar_off[r.rd], r.n, true))) ld = 0;
                }
                const unsigned long long m = __ballot(take);
                if (take) { const uint32_t d = begin + nv + lanes_below(m); voters[d] = r.rd; vld[d] = (uint32_t)ld; }
                nv += __popcll(m);
            }
        }
    } else
    for (uint32_t base = 0; base < np; base += 64) {
        uint32_t j = base + lane;
        bool take = false; uint32_t rd = NONE32; int ld = 0;
        if (j < np && j != best && (rd = side[begin + j]) != NONE32) {
            gce_core rk = b.core[rd]; const uint32_t *rc = b.cigar + b.cigar_off[rd];
            take = d_is_part_of(ocig, ok.n_cigar, rc, rk.n_cigar, left_mode);
            if (take) {
                ld = rk.l_qseq - ok.l_qseq;
                if (ld != 0 && rk.pos == ok.pos && d_is_part_of(ocig, ok.n_cigar, rc, rk.n_cigar, true)) ld = 0;
            }
        }
        unsigned long long m = __ballot(take);
        // NOTE: voters overwrite w.pl/w.pr at [begin+1 ...]; the source arrays here are gpl/gpr (distinct buffers)
        if (take) { uint32_t d = begin + nv + lanes_below(m); voters[d] = rd; vld[d] = (uint32_t)ld; }
        nv += __popcll(m);
    }
    WAVE_SYNC();
    // ---- Group::makeConsensus (group.cpp:320-579)
    int len = ok.l_qseq;
    if (ok.n_cigar == 0) {                                                    // group.cpp:354-360
        int mn = len;
        for (uint32_t q = lane; q < nv; q += 64) mn = min(mn, b.core[voters[begin + q]].l_qseq);
        len = wave_min(mn);
    }
    sp.out = out; sp.nv = nv; sp.left_mode = left_mode; sp.len = len; sp.voters = voters; sp.vld = vld;
    if (ok.isize != 0 && ok.tid >= 0 && ok.tid < p.n_ref) {                  // group.cpp:362-367 -> Reference::getData (reference.cpp:33-70)
        const uint8_t *rd = p.ref_data[ok.tid];
        int64_t need_len = (int64_t)d_ref_offset(ocig, ok.n_cigar, len - 1) + 1;
        if (rd && (int64_t)ok.pos + need_len < p.ref_len[ok.tid]) { sp.ref = rd; sp.ref_len = p.ref_len[ok.tid]; }
    }
    return sp;
}

// Group::consensusMergeBam for one side (group.cpp:136-318), wave-level: side_prepare + the column votes.  Returns the template
// read or NONE32.
__device__ uint32_t side_consensus(const DevBatch &b, const DevParams &p, const Work &w, uint32_t begin, uint32_t np, bool is_left,
                                   uint32_t *tally, int lane, int32_t *nm_slot) {
    const SidePrep sp = side_prepare(b, p, w, begin, np, is_left, lane);
    if (sp.out == NONE32) return NONE32;
    const uint32_t out = sp.out, nv = sp.nv; const int len = sp.len;
    const gce_core ok = b.core[out];
    const uint32_t *ocig = b.cigar + b.cigar_off[out];
    VoteCtx v;
    v.b = &b; v.p = &p; v.w = &w; v.out = out; v.nv = nv; v.vbase = begin; v.left_mode = sp.left_mode; v.len = len;
    v.ref = sp.ref; v.ref_len = sp.ref_len; v.ocig = ocig; v.oncig = ok.n_cigar; v.opos = ok.pos; v.tally = tally; v.voters = sp.voters; v.vld = sp.vld;
    uint8_t *oseq = b.seq + b.seq_off[out], *oqual = b.qual + b.qual_off[out];
    const int nbytes = (len + 1) >> 1;
    int minc = 0;
    if (nbytes <= 256) {
        // fast path: every lane owns bytes lane, lane+64, ... (2 columns each); results buffered in registers so the
        // `mismatchInc > 5` restore (group.cpp:537-558) is simply "do not write"
        uint8_t nb[4]; uint8_t nq0[4], nq1[4]; bool act0[4], act1[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            int bi = it * 64 + lane, c0 = bi * 2, c1 = c0 + 1;
            act0[it] = c0 < len; act1[it] = c1 < len;
            nb[it] = 0; nq0[it] = 0; nq1[it] = 0;
            if (it * 64 < nbytes) {           // wave-uniform
                uint8_t ob = act0[it] ? oseq[bi] : 0;
                int hi = ob >> 4, lo = ob & 0xF;
                { ColResult r = vote_column(v, c0, hi, lane, act0[it]); if (act0[it]) { hi = r.base; nq0[it] = (uint8_t)r.qual; minc += r.minc; } }
                { ColResult r = vote_column(v, c1, lo, lane, act1[it]); if (act1[it]) { lo = r.base; nq1[it] = (uint8_t)r.qual; minc += r.minc; } }
                nb[it] = (uint8_t)((hi << 4) | lo);
            }
        }
        minc = wave_sum(minc);
        bool restore = false;
        if (minc != 0) {                                                      // group.cpp:528-573
            if (b.nm_type[out] == 0) { if (lane == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
            else if (minc > 5) restore = true;
            else if (lane == 0) { int nn = b.nm[out] + minc; if (b.nm_type[out] == 'C' && nn >= 0 && nn <= 255) *nm_slot = nn; }
        }
        if (!restore) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                int bi = it * 64 + lane;
                if (act0[it]) { oseq[bi] = nb[it]; oqual[bi * 2] = nq0[it]; }
                if (act1[it]) oqual[bi * 2 + 1] = nq1[it];
            }
        }
    } else {
        // long templates: pass 1 counts mismatchInc without writing, pass 2 recomputes and writes
        for (int bb = 0; bb < nbytes; bb += 64) {                             // (wave-uniform trip count: vote_column broadcasts)
            const int bi = bb + lane; const bool in = bi < nbytes;
            uint8_t ob = in ? oseq[bi] : 0;
            int c0 = bi * 2, c1 = c0 + 1;
            { ColResult r = vote_column(v, c0, ob >> 4, lane, in && c0 < len); if (in && c0 < len) minc += r.minc; }
            { ColResult r = vote_column(v, c1, ob & 0xF, lane, in && c1 < len); if (in && c1 < len) minc += r.minc; }
        }
        minc = wave_sum(minc);
        bool restore = false;
        if (minc != 0) {
            if (b.nm_type[out] == 0) { if (lane == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
            else if (minc > 5) restore = true;
            else if (lane == 0) { int nn = b.nm[out] + minc; if (b.nm_type[out] == 'C' && nn >= 0 && nn <= 255) *nm_slot = nn; }
        }
        if (!restore) {
            for (int bb = 0; bb < nbytes; bb += 64) {
                const int bi = bb + lane; const bool in = bi < nbytes;
                uint8_t ob = in ? oseq[bi] : 0;
                int c0 = bi * 2, c1 = c0 + 1, hi = ob >> 4, lo = ob & 0xF;
                { ColResult r = vote_column(v, c0, hi, lane, in && c0 < len); if (in && c0 < len) { hi = r.base; oqual[c0] = (uint8_t)r.qual; } }
                { ColResult r = vote_column(v, c1, lo, lane, in && c1 < len); if (in && c1 < len) { lo = r.base; oqual[c1] = (uint8_t)r.qual; } }
                if (in) oseq[bi] = (uint8_t)((hi << 4) | lo);
            }
        }
    }
    WAVE_SYNC();
    return out;
}

// UMI of a consensus record: MI tag of the record itself if it has one, else parsed from the qname it now carries
// (BamUtil::getUMI, bamutil.cpp:23-38, after BamUtil::copyQName).
__device__ inline void d_record_umi(const DevBatch &b, const DevParams &p, const Work &w, uint32_t rec, uint32_t name_src,
                                    const char *&u, int &ul) {
    const uint64_t vr = w.uinfo[rec];
    if (uinfo_hm(vr)) { u = uinfo_ptr(b, vr); ul = uinfo_len(vr); return; }
    const uint64_t vn = name_src == rec ? vr : w.uinfo[name_src];
    if (!uinfo_hm(vn)) { u = uinfo_ptr(b, vn); ul = uinfo_len(vn); return; }
    const char *q = d_qname(b, name_src); int s0, l0;
    d_umi_slice(q, p, s0, l0);
    u = q + s0; ul = l0;
}

// ---- generic (any depth, any nibble, any length) consensus of deferred group sides: LDS tallies, global scratch
// The three argument blocks come through DEVICE MEMORY (round 5): side_consensus is a real call that takes them by reference, and by-value kernel arguments whose address
// is taken are copied into scratch in the kernel's prologue -- 992 bytes per lane, 124 MB written by every launch (512 blocks x 256 lanes) in front of the first look at the
// list, as a rule for a few hundred sides (profiles/r05_z_hbm_traffic.csv).
struct SlowArgs { DevBatch b; DevParams p; Work w; };
__global__ __launch_bounds__(256) void k_consensus_slow(const SlowArgs *a) {
    __shared__ uint32_t s_tally[WAVES_PER_BLOCK][48 * 64];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const Work &w = a->w;
    const uint32_t n_slow = (uint32_t)w.si->n_slow;
    for (uint32_t idx = blockIdx.x * WAVES_PER_BLOCK + wv; idx < n_slow; idx += gridDim.x * WAVES_PER_BLOCK) {
        uint32_t e = w.slow_list[idx], gi = e >> 1; bool is_left = !(e & 1);
        if (w.gen_flag[e] == 2) continue;                                       // finished by k_vote_deep (gce_deep.hpp)
        uint32_t c = w.gl_cluster[gi], g = gi - w.cl_gbase[c], cstart = w.cl_start[c];
        uint32_t begin = w.grp_begin[cstart + g], np = w.grp_n[cstart + g];
        uint32_t out = side_consensus(a->b, a->p, w, begin, np, is_left, s_tally[wv], lane, w.rp_nm + e);
        if (lane == 0) { if (is_left) w.rp_left[gi] = out; else w.rp_right[gi] = out; }
        WAVE_SYNC();
    }
}

// BamUtil::getRefOffset with the single-M case ("150M") answered from the first CIGAR word
__device__ __forceinline__ int ref_off_fast(const uint32_t *cig, int n, uint32_t c0, int qpos) {
    if (n == 1 && cig_op(c0) == 0) return qpos < cig_len(c0) ? qpos : -1;
    return d_ref_offset(cig, n, qpos);
}
// BamUtil::isPartOf with the single-op case (e.g. 150M vs 148M) answered from registers
__device__ __forceinline__ bool part_of_fast(uint32_t pc0, int pnc, const uint32_t *pcig, uint32_t wc0, int wnc, const uint32_t *wcig, bool left) {
    if (pnc == 0) return true;
    if (wnc < pnc) return false;
    if (pnc == 1) {                                      // the part's only op is also its last: it may be shorter (bamutil.cpp:233-236)
        const uint32_t wv = wnc == 1 ? wc0 : (left ? wcig[0] : wcig[wnc - 1]);
        return cig_op(pc0) == cig_op(wv) && cig_len(pc0) <= cig_len(wv);
    }
    return d_is_part_of(pcig, pnc, wcig, wnc, left);     // both CIGARs have >= 2 ops: their offsets were loaded
}

// (score, qual-sum) lexicographic order used by the top/second scans of group.cpp:394-416
__device__ __forceinline__ bool lex_gt(int s1, int q1, int s2, int q2) { return s1 > s2 || (s1 == s2 && q1 > q2); }
__device__ __forceinline__ bool lex_eq(int s1, int q1, int s2, int q2) { return s1 == s2 && q1 == q2; }

struct Tally5 { int cnt[5], ss[5], qs[5], tq[5]; int total; };
__device__ __forceinline__ void tally_clear(Tally5 &t) {
#pragma unroll
    for (int k = 0; k < 5; k++) { t.cnt[k] = 0; t.ss[k] = 0; t.qs[k] = 0; t.tq[k] = 0; }
    t.total = 0;
}
// returns false for a nibble outside {1,2,4,8,15} (handled by the generic kernel)
__device__ __forceinline__ bool tally_add(Tally5 &t, int nib, int q, int s) {
    int m0 = nib == 1, m1 = nib == 2, m2 = nib == 4, m3 = nib == 8, m4 = nib == 15;
    int m[5] = {m0, m1, m2, m3, m4};
#pragma unroll
    for (int k = 0; k < 5; k++) { t.cnt[k] += m[k]; t.ss[k] += m[k] * s; t.qs[k] += m[k] * q; t.tq[k] = max(t.tq[k], m[k] * q); }
    t.total += s;
    return (m0 | m1 | m2 | m3 | m4) != 0;
}

// One column of Group::makeConsensus (group.cpp:394-525) from the 5 real bins; the 11 other nibble bins are all
// (count 0, score 0, qual 0) and only enter through the `>=` later-bin tie rule (quirk Q6):
//   top    = LAST index whose (score, qualsum) equals the maximum over all 16 bins
//   second = the same over the bins other than top
struct ColOut { int base, qual, minc; };
__device__ __forceinline__ ColOut decide_column(const Tally5 &t, const DevParams &p, int out_base, int ref4) {
    const int IDX[5] = {1, 2, 4, 8, 15};
    int ms = 0, mq = 0;                                   // the absent bins contribute (0,0)
#pragma unroll
    for (int k = 0; k < 5; k++) if (lex_gt(t.ss[k], t.qs[k], ms, mq)) { ms = t.ss[k]; mq = t.qs[k]; }
    int top, tk = -1;                                     // tk: real-bin slot of top, -1 if top is an absent bin
    if (lex_eq(t.ss[4], t.qs[4], ms, mq)) { top = 15; tk = 4; }
    else if (ms == 0 && mq == 0) { top = 14; }
    else {
        top = 1; tk = 0;
#pragma unroll
        for (int k = 3; k >= 1; k--) if (tk == 0 && lex_eq(t.ss[k], t.qs[k], ms, mq)) { top = IDX[k]; tk = k; }
        // (descending scan: the highest index with the maximum wins; slot 0 is the fallback)
    }
    int top_s = ms;
    int s2 = 0, q2 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) if (k != tk && lex_gt(t.ss[k], t.qs[k], s2, q2)) { s2 = t.ss[k]; q2 = t.qs[k]; }
    int sk = -1;
    if (tk != 4 && lex_eq(t.ss[4], t.qs[4], s2, q2)) sk = 4;
    else if (s2 == 0 && q2 == 0) sk = -1;                 // an absent bin (14, or 13 when top is 14): count 0, quals 0
    else {
#pragma unroll
        for (int k = 3; k >= 0; k--) if (sk < 0 && k != tk && lex_eq(t.ss[k], t.qs[k], s2, q2)) sk = k;
    }
    int top_num = 0, top_qual = 0, sec_num = 0, sec_q = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { if (k == tk) { top_num = t.cnt[k]; top_qual = t.tq[k]; } if (k == sk) { sec_num = t.cnt[k]; sec_q = t.qs[k]; } }
    ColOut r; r.base = out_base; r.minc = 0;
    bool need = false;
    if (sec_num == 0) {                                                       // group.cpp:421-428
        if (top_s >= p.base_score_req && top_qual >= p.moderate_q) { r.qual = top_qual; return r; }
        need = true;
    }
    if (sec_num == 1) {                                                       // group.cpp:442-457
        if (sec_q <= p.low_q) { if (top_num < 2 && top_qual < p.high_q) need = true; }
        else { if (top_num < 3 || top_qual < p.high_q) need = true; }
    }
    if (sec_num > 1 && ((double)top_s < p.score_percent_req * (double)t.total || top_qual < p.moderate_q)) need = true;   // :460-464
    if (top_s < p.base_score_req || top_qual <= p.low_q) need = true;         // :466-467
    if (need && ref4 != 0) {                                                  // :470-501 (ref4 is one of 1,2,4,8)
        int rk = ref4 == 1 ? 0 : ref4 == 2 ? 1 : ref4 == 4 ? 2 : 3;
        int rbq = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k == rk) rbq = t.tq[k];              // highest quality among the ref-consistent voters (< 128 here)
        if (rbq >= p.high_q) top = ref4;
        if (top_qual < p.moderate_q) top = ref4;
        if (top == ref4) top_qual = rbq;
    }
    if (out_base != top) {                                                    // :503-524
        r.base = top;
        if (ref4 != 0) { if (out_base == ref4) r.minc = 1; else if (top == ref4) r.minc = -1; }
    }
    r.qual = top_qual;
    return r;
}

// decide_column with the (score, qual-sum, bin index) triple of every bin packed into one 32-bit key, so that "last bin with
// the lexicographic maximum" is a plain unsigned max:  key = (score + off) << 17 | qualsum << 4 | index.
//   off = voters x score_bias makes every score sum non-negative;  <= 64 voters, score_max + bias <= 255, quals < 128
//   (callers send anything else to the generic kernel)  =>  14 + 13 + 4 bits.
// The eleven absent bins (0, 0) are represented by their last member: index 14, or 13 once 14 itself is the top.
__device__ __forceinline__ ColOut decide_column_packed(const Tally5 &t, const DevParams &p, int out_base, int ref4) {
    const int IDX[5] = {1, 2, 4, 8, 15};
    const int off = (t.cnt[0] + t.cnt[1] + t.cnt[2] + t.cnt[3] + t.cnt[4]) * p.score_bias;
    uint32_t key[5];
#pragma unroll
    for (int k = 0; k < 5; k++) key[k] = ((uint32_t)(t.ss[k] + off) << 17) | ((uint32_t)t.qs[k] << 4) | (uint32_t)IDX[k];
    const uint32_t kabs = ((uint32_t)off << 17) | 14u;
    const uint32_t tk = max(max(max(key[0], key[1]), max(key[2], key[3])), max(key[4], kabs));
    int top = (int)(tk & 15u);
    uint32_t sk = top == 14 ? kabs - 1u : kabs;
#pragma unroll
    for (int k = 0; k < 5; k++) sk = max(sk, IDX[k] == top ? 0u : key[k]);
    const int sec = (int)(sk & 15u);
    const int top_s = (int)(tk >> 17) - off, sec_q = (int)((sk >> 4) & 0x1FFFu);
    int top_num = 0, top_qual = 0, sec_num = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { if (IDX[k] == top) { top_num = t.cnt[k]; top_qual = t.tq[k]; } if (IDX[k] == sec) sec_num = t.cnt[k]; }
    ColOut r; r.base = out_base; r.minc = 0;
    bool need = false;
    if (sec_num == 0) {                                                       // group.cpp:421-428
        if (top_s >= p.base_score_req && top_qual >= p.moderate_q) { r.qual = top_qual; return r; }
        need = true;
    }
    if (sec_num == 1) {                                                       // group.cpp:442-457
        if (sec_q <= p.low_q) { if (top_num < 2 && top_qual < p.high_q) need = true; }
        else { if (top_num < 3 || top_qual < p.high_q) need = true; }
    }
    if (sec_num > 1 && ((double)top_s < p.score_percent_req * (double)t.total || top_qual < p.moderate_q)) need = true;   // :460-464
    if (top_s < p.base_score_req || top_qual <= p.low_q) need = true;         // :466-467
    if (need && ref4 != 0) {                                                  // :470-501 (ref4 is one of 1,2,4,8)
        int rbq = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (IDX[k] == ref4) rbq = t.tq[k];        // highest quality among the ref-consistent voters (< 128 here)
        if (rbq >= p.high_q) top = ref4;
        if (top_qual < p.moderate_q) top = ref4;
        if (top == ref4) top_qual = rbq;
    }
    if (out_base != top) {                                                    // :503-524
        r.base = top;
        if (ref4 != 0) { if (out_base == ref4) r.minc = 1; else if (top == ref4) r.minc = -1; }
    }
    r.qual = top_qual;
    return r;
}

typedef uint16_t u16_unaligned __attribute__((aligned(1)));

// One wave per (group, side): Group::consensusMergeBam + makeConsensus for groups of <= 64 pairs with register-resident
// pair metadata and register tallies.  Anything else is appended to slow_list for k_consensus_slow.
// This is the kernel behind the group kernel (gce_vote.hpp): it takes the group sides that one flags (gen_flag -> gen_list).
__device__ void consensus_fast_side(const DevBatch &b, const DevParams &p, const Work &w, uint32_t gi, bool is_left, uint8_t *s_res_wave, int lane) {
    const uint32_t begin = w.g_begin[gi], np = w.g_np[gi];
    uint32_t *rp_out = is_left ? w.rp_left : w.rp_right;
    // deep sides (> 64 pairs, or beyond the low-complexity threshold) are not this kernel's: k_vote put them on slow_list itself when it handed their
    // group on (gce_vote.hpp, P0 -- the same test), so that k_deep_prepare can start behind k_vote instead of behind a pass of this kernel over gen_list
    const bool deep_side = !(np == 1 && w.gpr[begin] == NONE32) && (np > 64 || (int)np > p.skip_low_complexity_thr);
    if (deep_side) return;
    if (np == 1 && w.gpr[begin] == NONE32) {                                  // group.cpp:73-77: returned untouched
        if (lane == 0) rp_out[gi] = is_left ? w.gpl[begin] : NONE32;
        return;
    }
    const uint32_t *side = is_left ? w.gpl : w.gpr;
    // ---- per-lane pair metadata
    uint32_t rd = lane < (int)np ? side[begin + lane] : NONE32;
    const bool has = rd != NONE32;
    int pos = 0, lq = 0, nc = 0, rrp = 0; uint32_t c0 = 0; uint64_t cigo = 0, so = 0, qo = 0;
    int isz = 0; uint32_t patch = 0; int rtid = 0;
    if (has) {
        patch = w.spatch[rd];
        const ReadDesc k = load_desc(w.rdesc, rd);
        pos = k.pos; lq = k.lq; nc = k.nc; isz = k.isize; so = k.so; qo = k.qo; c0 = k.c0; rrp = pos + (k.rlen != RLEN_WALK ? k.rlen : d_cigar_rlen(b.cigar + b.cigar_off[rd], k.nc)); rtid = k.tid;
        if ((nc > 1 || (nc == 1 && cig_op(c0) != 0))) cigo = b.cigar_off[rd];     // anything but a single M block is walked from memory (rare)
    }
    const unsigned long long hmask = __ballot(has);
    if (!hmask) { if (lane == 0) rp_out[gi] = NONE32; return; }              // no read on this side: "no majority" / out == NULL
    // ---- leftReadMode (group.cpp:177-194)
    bool left_mode = is_left;
    if (!is_left) {
        int p0 = rl32(pos, __ffsll((long long)hmask) - 1);
        if (!__any(has && pos != p0)) left_mode = true;
    }
    // ---- containedBy (group.cpp:196-233) and the template pick (group.cpp:235-261).
    // Shortcut: when every read of the side carries the same single-op CIGAR and the same length (the usual "150M" group) and
    // the right-end filter cannot split them, every read is part of every other: containedBy = #reads for all, and the
    // (max containedBy, shorter, first in qname order) winner is simply the first read present.
    const int first_has = __ffsll((long long)hmask) - 1;
    const uint32_t c0f = (uint32_t)rl32((int)c0, first_has); const int lqf = rl32(lq, first_has);
    const bool uniform = !__any(has && (nc != 1 || c0 != c0f || lq != lqf)) && (is_left || left_mode);
    int best, bc;
    if (uniform) { best = first_has; bc = __popcll(hmask); }
    else {
        int cb = has ? 1 : 0;
        for (unsigned long long m = hmask; m; m &= m - 1) {
            const int j = __ffsll((long long)m) - 1;
            const int wnc = rl32(nc, j), wrrp = rl32(rrp, j); const uint32_t wc0 = (uint32_t)rl32((int)c0, j); const uint64_t wcig = rl64(cigo, j);
            if (has && lane != j && (is_left || rrp == wrrp) && part_of_fast(c0, nc, b.cigar + cigo, wc0, wnc, b.cigar + wcig, left_mode)) cb++;
        }
        int bl = has ? lq : 0;
        best = lane < (int)np ? lane : 0x7FFFFFFF; bc = lane < (int)np ? cb : -1;
        for (int o = 32; o > 0; o >>= 1) {
            int ob = __shfl_xor(best, o), oc = __shfl_xor(bc, o), ol = __shfl_xor(bl, o);
            bool better = oc > bc || (oc == bc && (ol < bl || (ol == bl && ob < best)));
            if (better) { best = ob; bc = oc; bl = ol; }
        }
    }
    if ((double)bc < (double)np * 0.4 && np != 1) { if (lane == 0) rp_out[gi] = NONE32; return; }        // group.cpp:264-266
    const uint32_t out = (uint32_t)rl32((int)rd, best);
    if (out == NONE32) { if (lane == 0) rp_out[gi] = NONE32; return; }
    const int o_pos = rl32(pos, best), o_lq = rl32(lq, best), o_nc = rl32(nc, best); const uint32_t o_c0 = (uint32_t)rl32((int)c0, best);
    const uint64_t o_cigo = rl64(cigo, best), o_so = rl64(so, best), o_qo = rl64(qo, best);
    const uint32_t *ocig = b.cigar + o_cigo;
    // ---- voters and lenDiff (group.cpp:287-313,339-348)
    bool take = false; int ld = 0;
    if (uniform) take = has;                                          // identical CIGARs: every read votes, lenDiff 0
    else if (has) {
        take = lane == best || part_of_fast(o_c0, o_nc, ocig, c0, nc, b.cigar + cigo, left_mode);
        if (take) { ld = lq - o_lq; if (ld != 0 && pos == o_pos && part_of_fast(o_c0, o_nc, ocig, c0, nc, b.cigar + cigo, true)) ld = 0; }
    }
    const unsigned long long vmask = __ballot(take);
    int len = o_lq;
    if (o_nc == 0) len = wave_min(take ? lq : 0x7FFFFFFF);           // group.cpp:354-360
    const int nbytes = (len + 1) >> 1;
    if (nbytes > 256) {                                                       // very long template: generic kernel
        if (lane == 0) w.slow_list[atomicAdd(&w.si->n_slow, 1u)] = gi * 2 + (is_left ? 0 : 1);
        return;
    }
    const int o_isz = rl32(isz, best), o_tid = rl32(rtid, best);
    // NM of the template (group.cpp:528-573), needed only at the very end: fetched now, off the critical path
    const int o_nm_type = b.nm_type[out], o_nm = b.nm[out];
    const uint8_t *ref = nullptr; int64_t ref_len = 0;
    if (o_isz != 0 && o_tid >= 0 && o_tid < p.n_ref) {                       // group.cpp:362-367 -> Reference::getData
        const uint8_t *rdp = p.ref_data[o_tid];
        int64_t need_len = (int64_t)ref_off_fast(ocig, o_nc, o_c0, len - 1) + 1;
        if (rdp && (int64_t)o_pos + need_len < p.ref_len[o_tid]) { ref = rdp; ref_len = p.ref_len[o_tid]; }
    }
    uint8_t *oseq = b.seq + o_so, *oqual = b.qual + o_qo;
    uint8_t *resb = s_res_wave, *resq = s_res_wave + 512;
    uint16_t *cplx = (uint16_t *)(s_res_wave + 1024);
    // ---- pass A: every column, 5 accumulators.  A column whose voters all show the same A/C/G/T/N nibble with
    //      total score >= baseScoreReq (> 0) and top quality >= moderate takes group.cpp:421-428's early accept:
    //      base unchanged, qual = max qual.  Everything else is queued for the full 16-bin rule cascade (pass B).
    const int accept_score = max(p.base_score_req, 1);
    int n_cplx = 0; bool odd = false;
    const bool even_ld = left_mode || !__any(take && (ld & 1));      // every voter's columns stay byte aligned
    const int nvot = __popcll(vmask);
    const bool swar_ok = even_ld && len <= 256 && nvot * (p.score_max + p.score_bias) <= 255 && p.q2s_swar_ok && accept_score + nvot * p.score_bias <= 255;
    if (swar_ok) {
        // SWAR form: one lane = 4 consecutive columns = 2 packed-base bytes + 4 quals + 4 scores, i.e. three loads per voter.
        //   unanimity  : XOR of the voter's two base bytes with the template's, OR-accumulated (a zero nibble = all agree)
        //   score sum  : packed byte add (scores are stored biased >= 0 and nv * max < 256, so bytes never carry)
        //   top quality: packed byte max (quals < 128; anything else is handed to the generic kernel)
        const int c4 = 4 * lane;
        const bool act = c4 < len;
        const int nval = act ? min(4, len - c4) : 0;
        const uint32_t nmask = nval >= 4 ? 0xFFFFu : nval == 3 ? 0xF0FFu : nval == 2 ? 0x00FFu : nval == 1 ? 0x00F0u : 0u;
        const uint32_t bmask = nval >= 4 ? 0xFFFFFFFFu : nval == 0 ? 0u : ((1u << (8 * nval)) - 1u);
        uint32_t t16 = 0;
        if (act) t16 = *(const u16_unaligned *)(oseq + 2 * lane);
        uint32_t dacc = 0, ssum = 0, tqm = 0, qor = 0, cnt = 0;
        const int s_min = min(min(p.s_high, p.s_moderate), min(p.s_low, p.s_bad));
        const bool lower_bound_ok = nvot * s_min >= accept_score;      // heuristic only: any lower bound keeps the result exact
        const uint32_t smin4 = 0x01010101u * (uint32_t)((s_min + p.score_bias) & 0xFF);
        unsigned long long m = vmask;
        while (m) {
            int vv[4]; uint32_t s16[4], q4[4], sc4[4], vm[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { vv[u] = m ? __ffsll((long long)m) - 1 : -1; if (m) m &= m - 1; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                s16[u] = t16; q4[u] = 0; sc4[u] = 0; vm[u] = 0;
                if (vv[u] >= 0) {                                              // wave-uniform
                    const int v = vv[u];
                    const uint64_t vso = rl64(so, v), vqo = rl64(qo, v); const int vld = left_mode ? 0 : rl32(ld, v), vlq = rl32(lq, v);
                    const uint32_t vpatch = (uint32_t)rl32((int)patch, v);
                    const int r0 = c4 + vld;
                    if (act) {
                        if (r0 >= 0 && r0 + nval <= vlq) {                     // the whole unit lies inside the voter
                            s16[u] = *(const u16_unaligned *)(b.seq + vso + (r0 >> 1));
                            q4[u] = *(const u32_unaligned *)(b.qual + vqo + r0);
                            // scores (vpatch is wave-uniform): qual2score of the four quals; a constant for a read scored without a
                            // usable mate; the stored bytes inside the voter's mate-overlap patch
                            // `lower_bound_ok` (wave-uniform): qual2score(q) >= min(s_*), and voters x min(s_*) already reaches
                            // baseScoreReq, so the constant lower bound decides the early accept exactly (true sum >= bound);
                            // columns that fail any test are recomputed exactly in pass B either way.
                            if (vpatch == 0u) sc4[u] = lower_bound_ok ? smin4 : d_q2s4_biased(p, q4[u]);
                            else if (vpatch == GCE_PATCH_CONST) sc4[u] = 0x01010101u * (uint32_t)((p.s_moderate + p.score_bias) & 0xFF);
                            else {
                                const int ps = (int)(vpatch & 0xFFFF), pe = ps + (int)(vpatch >> 16);
                                const int a_ = max(ps - r0, 0), z_ = min(pe - r0, 4);          // bytes [a_, z_) of the unit are patch
                                const uint32_t pmk = z_ > a_ ? ((z_ >= 4 ? 0xFFFFFFFFu : ((1u << (8 * z_)) - 1u)) & ~((1u << (8 * a_)) - 1u)) : 0u;
                                const uint32_t st4 = *(const u32_unaligned *)((const uint8_t *)w.score + vqo + r0);
                                sc4[u] = ((lower_bound_ok ? smin4 : d_q2s4_biased(p, q4[u])) & ~pmk) | (st4 & pmk);
                            }
                            vm[u] = bmask;
                        } else {                                               // unit straddles an end of the voter: byte by byte
                            uint32_t sx = t16;
                            for (int k = 0; k < nval; k++) {
                                const int rp = r0 + k;
                                if (rp >= 0 && rp < vlq) {
                                    const int nb = d_nib(b.seq + vso, rp), sh = 8 * (k >> 1) + ((k & 1) ? 0 : 4);
                                    sx = (sx & ~(0xFu << sh)) | ((uint32_t)nb << sh);
                                    q4[u] |= (uint32_t)b.qual[vqo + rp] << (8 * k);
                                    sc4[u] |= (uint32_t)((d_score_at(p, w.score + vqo, vpatch, rp, b.qual[vqo + rp]) + p.score_bias) & 0xFF) << (8 * k);
                                    vm[u] |= 0xFFu << (8 * k);
                                }
                            }
                            s16[u] = sx;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t qv = q4[u] & vm[u];
                dacc |= (s16[u] ^ t16) & nmask;
                ssum += sc4[u] & vm[u];
                cnt += vm[u] & 0x01010101u;
                qor |= qv;
                const uint32_t ge = ((tqm | 0x80808080u) - qv) & 0x80808080u;          // per byte: tqm >= qv
                const uint32_t sel = (ge - (ge >> 7)) | ge;                            // 0xFF where tqm >= qv
                tqm = (tqm & sel) | (qv & ~sel);
            }
        }
        if (qor & 0x80808080u) odd = true;
        bool cq[4];
        {
            // the four columns of the lane decided on packed bytes (byte k = column c4 + k; bit 7 of a byte = that column's flag)
            //   nibble pairs (hi = even column) -> one nibble per byte in column order: v_perm of the two masked halves
            const uint32_t tn4 = __builtin_amdgcn_perm(0u, ((t16 >> 4) & 0x0F0Fu) | ((t16 & 0x0F0Fu) << 16), 0x03010200u);
            const uint32_t dn4 = __builtin_amdgcn_perm(0u, ((dacc >> 4) & 0x0F0Fu) | ((dacc & 0x0F0Fu) << 16), 0x03010200u);
            const uint32_t differ = (dn4 + 0x7F7F7F7Fu) & 0x80808080u;                       // some voter shows another nibble
            //   template nibble in {1,2,4,8,15}: two 8-entry byte tables selected by bit 3
            const uint32_t sel = tn4 & 0x07070707u;
            const uint32_t v_lo = __builtin_amdgcn_perm(0x000000FFu, 0x00FFFF00u, sel);      // 1,2,4
            const uint32_t v_hi = __builtin_amdgcn_perm(0xFF000000u, 0x000000FFu, sel);      // 8,15
            const uint32_t hi8 = ((tn4 >> 3) & 0x01010101u) * 0xFFu;
            const uint32_t valid = (v_hi & hi8) | (v_lo & ~hi8);
            //   top quality >= moderate (bytes < 128, checked above)
            const uint32_t ge_q = ((tqm | 0x80808080u) - 0x01010101u * (uint32_t)p.moderate_q) & 0x80808080u;
            //   score sum >= baseScoreReq:  biased sum >= accept + count * bias, compared in two 16-bit halves per parity
            const uint32_t rhs = cnt * (uint32_t)p.score_bias + 0x01010101u * (uint32_t)accept_score;
            const uint32_t ge_e = (((ssum & 0x00FF00FFu) | 0x01000100u) - (rhs & 0x00FF00FFu)) & 0x01000100u;
            const uint32_t ge_o = ((((ssum >> 8) & 0x00FF00FFu) | 0x01000100u) - ((rhs >> 8) & 0x00FF00FFu)) & 0x01000100u;
            const uint32_t ge_s = (ge_e >> 1) | (ge_o << 7);
            const uint32_t contested = ~(ge_q & ge_s & ~differ & valid) & bmask & 0x80808080u;
            if (act) { *(uint32_t *)(resb + c4) = tn4; *(uint32_t *)(resq + c4) = tqm; }
#pragma unroll
            for (int k = 0; k < 4; k++) cq[k] = (contested >> (8 * k + 7)) & 1u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long mk = __ballot(cq[k]);
            if (cq[k]) cplx[n_cplx + lanes_below(mk)] = (uint16_t)(c4 + k);
            n_cplx += __popcll(mk);
        }
    } else
    for (int it = 0; it * 64 < nbytes; it++) {
        const int bi = it * 64 + lane, col0 = bi * 2;
        const bool a0 = col0 < len, a1 = col0 + 1 < len;
        uint32_t pm0 = 0, pm1 = 0; int ss0 = 0, ss1 = 0, tq0 = 0, tq1 = 0, qor = 0;
        for (unsigned long long m = vmask; m; m &= m - 1) {
            const int v = __ffsll((long long)m) - 1;
            const uint64_t vso = rl64(so, v), vqo = rl64(qo, v); const int vld = left_mode ? 0 : rl32(ld, v), vlq = rl32(lq, v);
            const int r0 = col0 + vld, r1 = r0 + 1;
            const uint8_t *vs = b.seq + vso; const uint8_t *vq = b.qual + vqo; const int8_t *vsc = w.score + vqo;
            const uint32_t vpatch = (uint32_t)rl32((int)patch, v);
            const bool in0 = a0 && r0 >= 0 && r0 < vlq, in1 = a1 && r1 >= 0 && r1 < vlq;
            if (in0 && in1 && !(r0 & 1)) {                                   // aligned pair of columns: one seq byte, 2+2 bytes
                uint8_t sb = vs[r0 >> 1];
                uint16_t qq = *(const u16_unaligned *)(vq + r0);
                int q0 = qq & 0xFF, q1 = qq >> 8;
                pm0 |= 1u << (sb >> 4); pm1 |= 1u << (sb & 0xF);
                ss0 += d_score_at(p, vsc, vpatch, r0, q0); ss1 += d_score_at(p, vsc, vpatch, r1, q1);
                tq0 = max(tq0, q0); tq1 = max(tq1, q1); qor |= q0 | q1;
            } else {
                if (in0) { int q0 = vq[r0]; pm0 |= 1u << d_nib(vs, r0); ss0 += d_score_at(p, vsc, vpatch, r0, q0); tq0 = max(tq0, q0); qor |= q0; }
                if (in1) { int q1 = vq[r1]; pm1 |= 1u << d_nib(vs, r1); ss1 += d_score_at(p, vsc, vpatch, r1, q1); tq1 = max(tq1, q1); qor |= q1; }
            }
        }
        if ((qor & 0x80) || ((pm0 | pm1) & ~0x8116u)) odd = true;            // qual >= 128 or a nibble outside {1,2,4,8,15}
        bool c0 = false, c1 = false;
        if (a0) {
            uint8_t ob = oseq[bi];
            resb[col0] = ob >> 4; resq[col0] = (uint8_t)tq0;
            c0 = !(__popc(pm0) == 1 && ss0 >= accept_score && tq0 >= p.moderate_q);
            if (a1) { resb[col0 + 1] = ob & 0xF; resq[col0 + 1] = (uint8_t)tq1; c1 = !(__popc(pm1) == 1 && ss1 >= accept_score && tq1 >= p.moderate_q); }
        }
        unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
        if (c0) cplx[n_cplx + lanes_below(m0)] = (uint16_t)col0;
        n_cplx += __popcll(m0);
        if (c1) cplx[n_cplx + lanes_below(m1)] = (uint16_t)(col0 + 1);
        n_cplx += __popcll(m1);
    }
    WAVE_SYNC();
    // ---- pass B: the contested columns.  Work items = (column, voter): every lane fetches ONE voter's (base, qual, score) for
    //      ONE contested column and adds it to that column's 5-bin tally in LDS (LDS atomics); then one lane per column runs
    //      the rule cascade + reference arbitration.  32 columns per round.
    int minc = 0;
    {
        uint32_t *tl = (uint32_t *)(s_res_wave + 2048);                     // [32 columns][5 bins][cnt, score, qualsum, topqual]
        uint8_t *vlist = s_res_wave + 2048 + 32 * 5 * 4 * 4;                // voter lanes in ascending order
        if (take) vlist[lanes_below(vmask)] = (uint8_t)lane;
        const uint32_t magic = ((1u << 20) + (uint32_t)nvot - 1) / (uint32_t)nvot;      // item / nvot == (item * magic) >> 20 for item < 2^11
        for (int cbase = 0; cbase < n_cplx; cbase += 32) {
            const int ncol = min(32, n_cplx - cbase);
            for (int k = lane; k < 32 * 5; k += 64) *(uint4 *)(tl + 4 * k) = make_uint4(0, 0, 0, 0);
            WAVE_SYNC();
            // the reference base of each contested column: requested before the voters' bytes so that both are in flight together
            int ref4 = 0;
            if (ref && lane < ncol) {
                const int ro = ref_off_fast(ocig, o_nc, o_c0, cplx[cbase + lane]);
                if (ro >= 0 && (int64_t)o_pos + ro < ref_len) ref4 = d_ref_nib(ref, (int64_t)o_pos + ro);
            }
            const int items = ncol * nvot;
            for (int ibase = 0; ibase < items; ibase += 64) {             // wave-uniform trip count: every lane executes the
                const int item = ibase + lane;                              // shuffles (an inactive source lane would read as 0)
                const bool live = item < items;
                const int it_ = live ? item : 0;
                const int c = (int)(((uint32_t)it_ * magic) >> 20), kx = it_ - c * nvot;
                const int vl = vlist[kx], col = cplx[cbase + c];
                const uint64_t vso = (uint64_t)__shfl((long long)so, vl), vqo = (uint64_t)__shfl((long long)qo, vl);
                const int vld = left_mode ? 0 : __shfl(ld, vl), vlq = __shfl(lq, vl);
                const uint32_t vpatch = (uint32_t)__shfl((int)patch, vl);
                const int rp = col + vld;
                if (live && rp >= 0 && rp < vlq) {
                    const int nb = d_nib(b.seq + vso, rp), q = b.qual[vqo + rp], sc = d_score_at(p, w.score + vqo, vpatch, rp, q);
                    const int bin = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 8 ? 3 : nb == 15 ? 4 : -1;
                    if (bin < 0 || (q & 0x80)) odd = true;
                    else {
                        uint32_t *t4 = tl + (c * 5 + bin) * 4;
                        atomicAdd(t4, 1u); atomicAdd(t4 + 1, (uint32_t)sc); atomicAdd(t4 + 2, (uint32_t)q); atomicMax(t4 + 3, (uint32_t)q);
                    }
                }
            }
            WAVE_SYNC();
            if (lane < ncol) {
                const int col = cplx[cbase + lane];
                Tally5 t; t.total = 0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint4 v4 = *(const uint4 *)(tl + (lane * 5 + k) * 4);
                    t.cnt[k] = (int)v4.x; t.ss[k] = (int)v4.y; t.qs[k] = (int)v4.z; t.tq[k] = (int)v4.w; t.total += (int)v4.y;
                }
                ColOut r = decide_column_packed(t, p, resb[col], ref4);
                resb[col] = (uint8_t)r.base; resq[col] = (uint8_t)r.qual; minc += r.minc;
            }
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
    if (__any(odd)) {                                                         // IUPAC nibble or qual >= 128 among the voters: generic kernel
        if (lane == 0) w.slow_list[atomicAdd(&w.si->n_slow, 1u)] = gi * 2 + (is_left ? 0 : 1);
        return;
    }
    minc = wave_sum(minc);
    bool restore = false;
    if (minc != 0) {                                                          // group.cpp:528-573
        if (o_nm_type == 0) { if (lane == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
        else if (minc > 5) restore = true;
        else if (lane == 0) { int nn = o_nm + minc; if (o_nm_type == 'C' && nn >= 0 && nn <= 255) w.rp_nm[gi * 2 + (is_left ? 0 : 1)] = nn; }
    }
    if (!restore) {
        for (int bi = lane; bi < nbytes; bi += 64) {
            const int c0 = 2 * bi;
            if (c0 + 1 < len) { oseq[bi] = (uint8_t)((resb[c0] << 4) | resb[c0 + 1]); *(u16_unaligned *)(oqual + c0) = (uint16_t)(resq[c0] | (resq[c0 + 1] << 8)); }
            else { oseq[bi] = (uint8_t)((resb[c0] << 4) | (oseq[bi] & 0xF)); oqual[c0] = resq[c0]; }
        }
    }
    if (lane == 0) rp_out[gi] = out;
}

// global-memory consensus, one wave per (group, side): the sides on gen_list (grid-stride, count read on the device)
__global__ __launch_bounds__(256, 5) void k_consensus_fast(DevBatch b, DevParams p, Work w) {
    __shared__ __attribute__((aligned(16))) uint8_t s_res[WAVES_PER_BLOCK][2048 + 2560 + 64];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    // the count only exists on the device: a capped grid strides over the list (a wave takes one or two entries when the list is
    // long, and leaves at once when it is short -- launching one block per POSSIBLE entry cost more than the work itself)
    const uint32_t n = (uint32_t)(w.si->hand_on >> 32);
    for (uint32_t idx = blockIdx.x * WAVES_PER_BLOCK + wv; idx < n; idx += gridDim.x * WAVES_PER_BLOCK) {
        const uint32_t e = w.gen_list[idx];
        consensus_fast_side(b, p, w, e >> 1, !(e & 1), s_res[wv], lane);
        WAVE_SYNC();
    }
}

// ===================================================================================================== finish
// Cluster::duplexMergeBam (cluster.cpp:199-244).  The reference loop is
//     for(i=0;i<len;i++){ if(seq1[i/2]==seq2[i/2]){ i++; continue; } compare base i; mismatch -> N / qual 0 in both }
// i.e. a two-phase automaton over the packed bytes: arriving at a byte on its EVEN index it examines both nibbles;
// after a masked high nibble whose low nibbles agree the bytes become equal, the `i++; continue` lands on the ODD index
// of the next byte, and while it stays on odd indices only whole-byte equality and LOW nibbles are examined (a high
// nibble mismatch is then skipped) until an unequal byte re-synchronises it.  The phase each byte is entered in is
// computed exactly from two ballots (EQ = bytes equal, TO = byte sends the walk to the odd phase).
// Only b1 (the surviving pair) is written: b2 is deleted right after the merge (cluster.cpp:151-152).
__device__ inline int d_duplex_merge_bam(const DevBatch &b, uint32_t r1, uint32_t r2, int lane) {
    int l1 = b.core[r1].l_qseq, l2 = b.core[r2].l_qseq;
    int len = min(l1, l2), diff = 0;
    uint8_t *s1 = b.seq + b.seq_off[r1], *q1 = b.qual + b.qual_off[r1];
    const uint8_t *s2 = b.seq + b.seq_off[r2];
    const int nb = (len + 1) >> 1;
    bool phase_odd = false;                                  // phase in which the chunk's first byte is entered (wave-uniform)
    for (int base = 0; base < nb; base += 64) {
        int k = base + lane;
        bool valid = k < nb;
        uint8_t x = valid ? s1[k] : 0, y = valid ? s2[k] : 0;
        int xh = x >> 4, xl = x & 0xF, yh = y >> 4, yl = y & 0xF;
        bool has_lo = (2 * k + 1) < len;
        bool ne = valid && x != y;
        bool mark_hi = ne && d_base_class(xh) != d_base_class(yh);
        bool to_odd = mark_hi && xl == yl && has_lo;          // bytes equal after masking -> the walk continues on odd indices
        unsigned long long EQ = __ballot(!ne), TO = __ballot(to_odd);
        unsigned long long oddin = 0;
        int pos = 0; bool st = phase_odd;
        while (pos < 64) {
            if (!st) {
                unsigned long long m = TO & (~0ull << pos);
                if (!m) { pos = 64; break; }
                pos = __ffsll((long long)m);                 // index + 1: the byte after the one that switched phase
                st = true;
            } else {
                unsigned long long m = ~EQ & (~0ull << pos);
                int nx = m ? __ffsll((long long)m) - 1 : 64; // first unequal byte met in odd phase (it re-synchronises)
                unsigned long long upto = nx >= 63 ? ~0ull : ((2ull << nx) - 1ull);
                oddin |= upto & (~0ull << pos);
                if (nx == 64) pos = 64; else { pos = nx + 1; st = false; }
            }
        }
        phase_odd = st;
        bool in_odd = (oddin >> lane) & 1ull;
        if (ne) {
            if (!in_odd) {
                if (mark_hi) { diff++; xh = 15; q1[2 * k] = 0; }
                if (has_lo && !(mark_hi && xl == yl) && d_base_class(xl) != d_base_class(yl)) { diff++; xl = 15; q1[2 * k + 1] = 0; }
            } else if (has_lo && d_base_class(xl) != d_base_class(yl)) { diff++; xl = 15; q1[2 * k + 1] = 0; }
            s1[k] = (uint8_t)((xh << 4) | xl);
        }
    }
    return wave_sum(diff) + (l1 > l2 ? l1 - l2 : l2 - l1);
}

__device__ inline void d_emit_pair(const Work &w, uint32_t gi, bool duplex) {      // Pair::writeSscsDcsTag + Gencore::outputPair
    uint32_t l = w.rp_left[gi], r = w.rp_right[gi];
    int fr = (int)min(w.rp_merge[gi], 65535u) & 0xFF;                               // low byte of an unsigned short (quirk Q8)
    int rr = (int)min(w.rp_rmerge[gi], 65535u) & 0xFF;
    OutRec o; o.fr = (int16_t)fr; o.rr = duplex ? (int16_t)rr : (int16_t)-1; o.pad = 0;
    if (l != NONE32) { w.out_flag[l] = 1; o.qname_src = w.rp_qsl[gi]; o.mate = r; o.nm_new = (int16_t)w.rp_nm[gi * 2]; w.orec[l] = o; }
    if (r != NONE32) { w.out_flag[r] = 1; o.qname_src = w.rp_qsr[gi]; o.mate = l; o.nm_new = (int16_t)w.rp_nm[gi * 2 + 1]; w.orec[r] = o; }
}

// The tail of Group::consensusMerge (group.cpp:104-132) — mMergeReads, qname reconciliation, the new Pair's UMI — one
// THREAD per group; groups of clusters that cannot form a duplex (no UMI, --no_duplex, or a single group) are also
// filtered, tagged and emitted here (cluster.cpp:169-183; with one group the duplex loop finds no partner, :157-166).
__global__ __launch_bounds__(256) void k_group_tail(DevBatch b, DevParams p, Work w, uint32_t n_groups) {
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_groups) return;
    const uint32_t c = w.gl_cluster[gi];
    const uint32_t begin = w.g_begin[gi], np = w.g_np[gi];
    uint32_t left = w.rp_left[gi], right = w.rp_right[gi];
    {   // the NM patches k_vote deferred (group.cpp:528-573): mismatchInc != 0 reads the template's NM tag -- absent: the reference crashes (quirk Q9);
        // mismatchInc > 5: the template was restored, NM stays; else NM + mismatchInc goes into the tag if it is stored as type 'C' and stays a byte
        const int2 nv = *reinterpret_cast<const int2 *>(w.rp_nm + 2 * (size_t)gi);
        if (NM_IS_DEFERRED(nv.x) || NM_IS_DEFERRED(nv.y)) {
            int2 o = nv;
            auto resolve = [&](int v, uint32_t out) -> int {
                if (!NM_IS_DEFERRED(v)) return v;
                const int minc = v - NM_DEFER, ty = b.nm_type[out], nn = b.nm[out] + minc;
                if (ty == 0) { raise_error(w.si, GCE_ERR_NM_MISSING, out); return -1; }
                return (minc <= 5 && ty == 'C' && nn >= 0 && nn <= 255) ? nn : -1;
            };
            o.x = resolve(nv.x, left); o.y = resolve(nv.y, right);
            *reinterpret_cast<int2 *>(w.rp_nm + 2 * (size_t)gi) = o;
        }
    }
    const uint8_t cflags = w.cl_hasumi[c];
    const uint32_t G = w.cl_ngroups[c];
    const bool single = np == 1 && w.gpr[begin] == NONE32;
    uint32_t qsl = left, qsr = right;                                     // BamUtil::copyQName sources (== the record itself if unchanged)
    // the two templates' name lengths and UMI places: one word each (round 5; core record + UMI pointer + UMI length + MI flag were four sectors per side)
    const uint64_t uL = left != NONE32 ? w.uinfo[left] : 0ull, uR = right != NONE32 ? w.uinfo[right] : 0ull;
    if (!single) {
        if (cflags & 2) {                                                 // cross-contig cluster: group.cpp:79-99,109-112
            uint32_t ntc = NONE32; int bl = 0;
            for (uint32_t k = 0; k < np; k++) {
                uint32_t l = w.gpl[begin + k];
                int ll = d_lqname_pad(b.core[l]);
                if (ntc == NONE32 || ll < bl || (ll == bl && d_strcmp(d_qname(b, l), d_qname(b, ntc)) < 0)) { ntc = l; bl = ll; }
            }
            if (left != NONE32 && ntc != NONE32 && ntc != left) {
                if (d_lqname_pad(b.core[left]) < d_lqname_pad(b.core[ntc])) raise_error(w.si, GCE_ERR_QNAME_SHORT, left);
                qsl = ntc;
            }
        } else if (left != NONE32 && right != NONE32) {                   // group.cpp:114-123
            if (uinfo_lqname_pad(uL) <= uinfo_lqname_pad(uR)) qsr = left;
            else qsl = right;
        }
    }
    w.rp_qsl[gi] = qsl; w.rp_qsr[gi] = qsr;
    const uint32_t merge = single ? 1 : np;
    const bool duplex_cluster = (cflags & 1) && !p.disable_duplex && G >= 2;
    const char *u = nullptr; int ul = 0;                                  // Pair::setLeft / setRight (pair.cpp:188-216)
    if ((cflags & 1) || b.mi) {
        auto rec_umi = [&](uint64_t vrec, uint32_t src, const char *&uo, int &ulo) {       // d_record_umi on the words already here
            if (uinfo_hm(vrec)) { uo = uinfo_ptr(b, vrec); ulo = uinfo_len(vrec); return; }
            const uint64_t vn = src == left ? uL : (src == right ? uR : w.uinfo[src]);
            if (!uinfo_hm(vn)) { uo = uinfo_ptr(b, vn); ulo = uinfo_len(vn); return; }
            const char *q = d_qname(b, src); int s0, l0;
            d_umi_slice(q, p, s0, l0);
            uo = q + s0; ulo = l0;
        };
        if (left != NONE32) rec_umi(uL, qsl, u, ul);
        if (right != NONE32) {
            const char *u2; int ul2;
            rec_umi(uR, qsr, u2, ul2);
            if (left != NONE32 && ul != 0) {                               // (two word loads per side instead of a byte loop with a data-dependent exit)
                bool same = ul == ul2;
                if (same && u == u2) { }                                    // both records carry the same read's name (the usual case): nothing to fetch
                else if (same && ul <= 24) { uint64_t a[3], c3[3]; load_be_words<3>(u, ul, a); load_be_words<3>(u2, ul2, c3); same = a[0] == c3[0] && a[1] == c3[1] && a[2] == c3[2]; }
                else if (same) same = d_bytes_equal(u, ul, u2, ul2);
                if (!same) raise_error(w.si, GCE_ERR_UMI_MISMATCH, right);
            }
            u = u2; ul = ul2;
        }
    }
    w.rp_merge[gi] = merge; w.rp_rmerge[gi] = 0; w.rp_umi[gi] = u; w.rp_umilen[gi] = (uint16_t)ul;
    if (duplex_cluster) { w.rp_state[gi] = RP_PENDING; w.rp_supp[gi] = -1; return; }
    const bool outp = !p.duplex_only && (int)merge >= p.cluster_size_req;
    w.rp_supp[gi] = (int)merge; w.rp_state[gi] = outp ? RP_OUT_SSCS : RP_DROPPED;
    if (outp) d_emit_pair(w, gi, false);
}

// Duplex matching (cluster.cpp:119-168), one wave per cluster that has UMIs and at least two groups.
// Round 5: a cluster of <= 64 groups keeps its groups in the LANES -- state, reads, merge count, and the two '_'-separated tokens of the UMI as zero-padded big-endian
// words (UMI characters are never NUL, so equal words are equal strings: Cluster::isDuplex, cluster.cpp:246-258, is two 64-bit compares) -- and the walk from the
// last group down is a broadcast and a ballot per group.  The walk over memory below (any number of groups, tokens of any length) loads every partner's UMI bytes
// and state again for every group: ~8 dependent round trips per group on a one-wave chain, 305 us at cfg5 for clusters of ~45 groups.
__device__ bool finish_cluster_lanes(const DevBatch &b, const DevParams &p, const Work &w, uint32_t g0, uint32_t G, int lane) {
    const bool in = lane < (int)G;
    const uint32_t gme = g0 + (uint32_t)lane;
    uint32_t l = NONE32, r = NONE32, m = 0; uint64_t t0 = 0, t1 = 0; bool two = false, lng = false;
    if (in) {
        const char *u = w.rp_umi[gme]; const int ul = w.rp_umilen[gme];
        l = w.rp_left[gme]; r = w.rp_right[gme]; m = w.rp_merge[gme];
        int s0, l0, s1, l1;
        two = d_split2(u, ul, s0, l0, s1, l1) == 2;
        if (two) {
            if (l0 > 8 || l1 > 8) lng = true;
            else { uint64_t a[1], c2[1]; load_be_words<1>(u + s0, l0, a); load_be_words<1>(u + s1, l1, c2); t0 = a[0]; t1 = c2[0]; }
        }
    }
    if (__any(lng)) return false;                            // a token beyond eight bytes: the walk over memory
    int st = RP_PENDING;                                     // (k_group_tail left every group of such a cluster pending)
    for (int idx = (int)G - 1; idx >= 0; idx--) {
        if (__shfl(st, idx) == RP_CONSUMED) continue;        // (wave-uniform)
        const uint64_t a0 = (uint64_t)__shfl((long long)t0, idx), a1 = (uint64_t)__shfl((long long)t1, idx);
        const int atwo = __shfl((int)two, idx);
        const bool hit = lane < idx && st == RP_PENDING && two && atwo && a0 == t1 && a1 == t0;
        const unsigned long long hm = __ballot(hit);
        const uint32_t gi = g0 + (uint32_t)idx;
        const uint32_t l1 = (uint32_t)__shfl((int)l, idx), r1 = (uint32_t)__shfl((int)r, idx), m1 = (uint32_t)__shfl((int)m, idx);
        if (hm) {
            const int found = __ffsll((long long)hm) - 1;    // the first partner in group order (cluster.cpp:130-141)
            const uint32_t g2 = g0 + (uint32_t)found;
            const uint32_t l2 = (uint32_t)__shfl((int)l, found), r2 = (uint32_t)__shfl((int)r, found), m2 = (uint32_t)__shfl((int)m, found);
            int diff = 0;
            if (l1 != NONE32 && l2 != NONE32) diff += d_duplex_merge_bam(b, l1, l2, lane);
            if (r1 != NONE32 && r2 != NONE32) diff += d_duplex_merge_bam(b, r1, r2, lane);
            const bool outp = diff <= p.duplex_mismatch_thr && (int)(m1 + m2) >= p.cluster_size_req;
            if (lane == idx) st = outp ? RP_OUT_DCS : RP_DROPPED;
            if (lane == found) st = RP_CONSUMED;
            if (lane == 0) {
                w.rp_supp[gi] = (int)(m1 + m2);
                w.rp_rmerge[gi] = m2;
                w.rp_state[gi] = outp ? RP_OUT_DCS : RP_DROPPED;
                w.rp_state[g2] = RP_CONSUMED;
                if (outp) d_emit_pair(w, gi, true);
            }
            WAVE_SYNC();                                     // (the merged bases / qualities of this pair before anybody reads them again)
        } else {
            const bool outp = !p.duplex_only && (int)m1 >= p.cluster_size_req;
            if (lane == idx) st = outp ? RP_OUT_SSCS : RP_DROPPED;
            if (lane == 0) { w.rp_supp[gi] = (int)m1; w.rp_state[gi] = outp ? RP_OUT_SSCS : RP_DROPPED; if (outp) d_emit_pair(w, gi, false); }
        }
    }
    return true;
}
__device__ void finish_cluster(const DevBatch &b, const DevParams &p, const Work &w, uint32_t c, int lane) {
    const uint32_t G = w.cl_ngroups[c];
    const uint32_t g0 = w.cl_gbase[c];
    if (G <= 64u && finish_cluster_lanes(b, p, w, g0, G, lane)) return;
    for (int idx = (int)G - 1; idx >= 0; idx--) {
        uint32_t gi = g0 + idx;
        if (w.rp_state[gi] == RP_CONSUMED) continue;
        const char *u1 = w.rp_umi[gi]; int ul1 = w.rp_umilen[gi];
        uint32_t found = NONE32;
        for (int base = 0; base < idx && found == NONE32; base += 64) {
            int i = base + lane;
            bool hit = i < idx && w.rp_state[g0 + i] == RP_PENDING && d_is_duplex(u1, ul1, w.rp_umi[g0 + i], w.rp_umilen[g0 + i]);
            unsigned long long m = __ballot(hit);
            if (m) found = base + (__ffsll((long long)m) - 1);
        }
        uint32_t l1 = w.rp_left[gi], r1 = w.rp_right[gi], m1 = w.rp_merge[gi];
        if (found != NONE32) {
            uint32_t g2 = g0 + found;
            uint32_t l2 = w.rp_left[g2], r2 = w.rp_right[g2], m2 = w.rp_merge[g2];
            int diff = 0;
            if (l1 != NONE32 && l2 != NONE32) diff += d_duplex_merge_bam(b, l1, l2, lane);
            if (r1 != NONE32 && r2 != NONE32) diff += d_duplex_merge_bam(b, r1, r2, lane);
            bool outp = diff <= p.duplex_mismatch_thr && (int)(m1 + m2) >= p.cluster_size_req;
            if (lane == 0) {
                w.rp_supp[gi] = (int)(m1 + m2);
                w.rp_rmerge[gi] = m2;
                w.rp_state[gi] = outp ? RP_OUT_DCS : RP_DROPPED;
                w.rp_state[g2] = RP_CONSUMED;
                if (outp) d_emit_pair(w, gi, true);
            }
        } else {
            bool outp = !p.duplex_only && (int)m1 >= p.cluster_size_req;
            if (lane == 0) { w.rp_supp[gi] = (int)m1; w.rp_state[gi] = outp ? RP_OUT_SSCS : RP_DROPPED; if (outp) d_emit_pair(w, gi, false); }
        }
        WAVE_SYNC();
    }
}
// Few clusters need the duplex stage (UMIs and >= 2 groups): they are flagged, compacted, and then get a wave each -- a deep amplicon
// cluster with a hundred UMI groups keeps its wave busy for a long time, and 64 neighbouring clusters behind one wave (the first
// version) serialised exactly those.
__global__ __launch_bounds__(256) void k_finish_screen(Work w, uint32_t n_clusters, uint8_t *flag) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_clusters) flag[c] = w.cl_ngroups[c] >= 2 && (w.cl_hasumi[c] & 1);
}
__global__ __launch_bounds__(256) void k_finish(DevBatch b, DevParams p, Work w, const uint32_t *list, const unsigned long long *list_n) {
    const int lane = lane_id();
    const uint32_t n = (uint32_t)*list_n;
    for (uint32_t k = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6); k < n; k += gridDim.x * WAVES_PER_BLOCK) {
        finish_cluster(b, p, w, list[k], lane);
        WAVE_SYNC();
    }
}

// ===================================================================================================== Stats
// All counters are additive (stats.h:47-65).  Block-local accumulation in LDS, one global atomic per non-zero counter.
// (Stats::addRead of the emitted records: k_out_meta, gce_output.hpp)
__global__ __launch_bounds__(256) void k_stats(DevBatch b, Work w, uint32_t n_clusters, uint32_t n_groups) {
    __shared__ unsigned long long s_pre[GCE_STATS_WORDS], s_post[GCE_STATS_WORDS], s_np;
    for (int k = threadIdx.x; k < GCE_STATS_WORDS; k += blockDim.x) { s_pre[k] = 0; s_post[k] = 0; }
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    const uint64_t tid0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    // indices into gce_stats: 0 reads, 1 bases, 2 reads_unmapped, 3 bases_unmapped, 4 mismatches, 5 reads_with_mismatches,
    // 6 clusters, 7 multi, 8 molecules, 9 se, 10 pe, 11 sscs, 12 dcs, 13 uncounted, 14.. hist.
    // The scalar counters are summed in registers and reduced per wave at the end (64 lanes adding to one LDS word serialise);
    // only the histogram goes through LDS atomics.
    long long pre[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, post[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, np = 0, post_h1 = 0;
    for (uint64_t c = tid0; c < n_clusters; c += stride) {
        uint32_t G = w.cl_ngroups[c];
        if (G == 0) continue;
        pre[6] += 1; if (G > 1) pre[7] += 1;                                             // cluster.cpp:102
        np += w.cl_npairs[c];
        uint32_t nr = 0; { const uint32_t g0 = w.cl_gbase[c]; for (uint32_t g = 0; g < G; g++) { uint8_t st = w.rp_state[g0 + g]; nr += (st == RP_OUT_SSCS || st == RP_OUT_DCS); } }
        if (nr > 0) { post[6] += 1; if (nr > 1) post[7] += 1; }                         // cluster.cpp:185-187
    }
    for (uint64_t g = tid0; g < n_groups; g += stride) {
        int supp = w.rp_supp[g];
        if (supp < 0) continue;                                                          // consumed as the reverse strand: no event
        bool pe = w.rp_left[g] != NONE32 && w.rp_right[g] != NONE32;
        pre[8] += 1;                                                                     // Stats::addMolecule, stats.cpp:123-133
        if (supp < GCE_MAX_SUPPORTING_READS) atomicAdd(&s_pre[14 + supp], 1ull); else pre[13] += 1;
        pre[pe ? 10 : 9] += 1;
        uint8_t st = w.rp_state[g];
        if (st == RP_OUT_SSCS || st == RP_OUT_DCS) {
            post[st == RP_OUT_DCS ? 12 : 11] += 1;                                       // addSSCS / addDCS
            post[8] += 1; post_h1 += 1; post[pe ? 10 : 9] += 1;                          // outputPair: addMolecule(1, PE)
        }
    }
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const long long a = wave_sum64(pre[k]), c2 = wave_sum64(post[k]);
        if (lane == 0) { if (a) atomicAdd(&s_pre[k], (unsigned long long)a); if (c2) atomicAdd(&s_post[k], (unsigned long long)c2); }
    }
    { const long long a = wave_sum64(np), c2 = wave_sum64(post_h1); if (lane == 0) { if (a) atomicAdd(&s_np, (unsigned long long)a); if (c2) atomicAdd(&s_post[14 + 1], (unsigned long long)c2); } }
    __syncthreads();
    for (int k = threadIdx.x; k < GCE_STATS_WORDS; k += blockDim.x) {
        if (s_pre[k]) atomicAdd((unsigned long long *)&w.si->pre[k], s_pre[k]);
        if (s_post[k]) atomicAdd((unsigned long long *)&w.si->post[k], s_post[k]);
    }
    if (threadIdx.x == 0 && s_np) atomicAdd(&w.si->n_pairs_total, s_np);
}

// out_index: ascending list of emitted reads (3-phase scan over out_flag != 0)
__global__ __launch_bounds__(256) void k_flag_reduce(const uint8_t *flag, uint64_t n, uint64_t *part) {
    __shared__ uint64_t s[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE, v = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) { uint64_t i = base + k * 256 + threadIdx.x; v += (i < n && flag[i]) ? 1 : 0; }
    v = (uint64_t)wave_sum64((long long)v);
    if (lane_id() == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s[0] + s[1] + s[2] + s[3]) << 32;    // keep the total in the high word like the table scan
}
__global__ __launch_bounds__(256) void k_flag_apply(const uint8_t *flag, uint64_t n, const uint64_t *part, uint32_t *out_index) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = (uint32_t)(part[blockIdx.x] >> 32);
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        uint64_t i = base + k * 256 + threadIdx.x;
        bool f = i < n && flag[i];
        unsigned long long m = __ballot(f);
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        uint32_t woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        uint32_t carry = s_carry;
        if (f) out_index[carry + woff + lanes_below(m)] = (uint32_t)i;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}
