// gce_pair2.hpp — mate pairing + UMI grouping with SEVERAL clusters per wave: 16 lanes per cluster (four clusters per wave) for
// clusters of <= 16 reads, then 32 lanes per cluster (two per wave) for the ones that pass flagged on, then k_pairing_fast.
//
// k_pairing_fast is VALU-bound with 13 of 64 lanes carrying a read at the benchmark's depth; running two clusters as the two
// halves of a wave executes the same instruction stream once for both.  Same algorithm as k_pairing_fast (gce_kernels.hpp):
// names as big-endian words, hash-filtered name classes verified exactly, first / last read of a name = mLeft / mRight
// (pair.cpp:188-216), lexicographic rank against the first read of every other name (std::map order, cluster.cpp:260-273),
// the setRight UMI check, read -> pair transposition by ds_permute, greedy UMI grouping (cluster.cpp:57-100), group-contiguous
// layout.  What was wave-uniform is half-uniform here: masks are 32-bit per half, broadcasts are ds_bpermute from (half base
// + index), loops run to the larger half's trip count under per-lane predicates.
//
// Scope: <= SUB reads per cluster, names <= 64 bytes, UMIs <= 24 bytes.  Anything else is flagged and taken by the next wider
// instantiation / k_pairing_fast / the generic kernel in a later launch over the compacted list ("half" in the comments below
// = the SUB-lane group of a cluster).
#pragma once

template <int SUB> __device__ __forceinline__ int sub_max(int v) {       // (all lanes of the wave must call it)
    v = row_max16(v);
    if (SUB == 32) { const int t = __shfl_xor(v, 16); v = t > v ? t : v; }
    return v;
}
// my sub-group's bits of a wave ballot
template <int SUB> __device__ __forceinline__ uint32_t sub_ballot(bool pr, int hb) {
    const unsigned long long m = __ballot(pr);
    return (uint32_t)(m >> hb) & (SUB == 32 ? 0xFFFFFFFFu : ((1u << (SUB & 31)) - 1u));
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) { return (uint64_t)__shfl((long long)v, src); }

// Size classes in front of the pairing kernels: the quarter-wave kernel used to run over ALL clusters and flag the ones of more than 16 reads for the
// next one -- 45 % of cfg3's clusters, a dead quarter of a wave each, in a kernel that executes the same instruction stream whatever share of its lanes
// is alive.  Now it gets the compacted list of the clusters it can take; the others are flagged for the half-wave kernel here.
// Round 5: the three lists (<= 16 reads, <= 32, more) come out of ONE compaction over cl_n -- counts of the first two classes in the halves of one 64-bit partial per tile
// (k_scan_partials scans both at once; the third follows from the tile's place) -- instead of a flag pass and three flag compactions: 3 launches where there were 10.
__device__ __forceinline__ int pair_class(uint32_t n) { return n <= 16u ? 0 : (n <= 32u ? 1 : 2); }        // (beyond 64 reads: the full-wave kernel hands the cluster on itself)
__global__ __launch_bounds__(256) void k_pair_class_reduce(const uint32_t *cl_n, uint32_t n_clusters, uint64_t *part) {
    __shared__ uint64_t s[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t v = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        const uint64_t i = base + k * 256 + threadIdx.x;
        if (i < n_clusters) { const int c = pair_class(cl_n[i]); v += c == 0 ? (1ull << 32) : (c == 1 ? 1ull : 0ull); }
    }
    v = (uint64_t)wave_sum64((long long)v);
    if (lane_id() == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k_pair_class_apply(const uint32_t *cl_n, uint32_t n_clusters, const uint64_t *part, uint32_t *l16, uint32_t *l32, uint32_t *l64, StreamInfo *si) {
    __shared__ uint32_t s_w[4][2];
    __shared__ uint32_t s_c0, s_c1;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (threadIdx.x == 0) { s_c0 = (uint32_t)(part[blockIdx.x] >> 32); s_c1 = (uint32_t)part[blockIdx.x]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) si->n_pf_items = (unsigned long long)n_clusters - si->n_p16_items - si->n_pq_items;     // (k_scan_partials left the other two totals)
    __syncthreads();
    for (int k = 0; k < SCAN_TILE / 256; k++) {
        const uint64_t i = base + k * 256 + threadIdx.x;
        const int c = i < n_clusters ? pair_class(cl_n[i]) : -1;
        const unsigned long long m0 = __ballot(c == 0), m1 = __ballot(c == 1);
        if (lane == 0) { s_w[wv][0] = (uint32_t)__popcll(m0); s_w[wv][1] = (uint32_t)__popcll(m1); }
        __syncthreads();
        uint32_t o0 = s_c0, o1 = s_c1, t0 = 0, t1 = 0;
        for (int q = 0; q < 4; q++) { if (q < wv) { o0 += s_w[q][0]; o1 += s_w[q][1]; } t0 += s_w[q][0]; t1 += s_w[q][1]; }
        const uint32_t r0 = o0 + (uint32_t)lanes_below(m0), r1 = o1 + (uint32_t)lanes_below(m1);
        if (c == 0) l16[r0] = (uint32_t)i;
        else if (c == 1) l32[r1] = (uint32_t)i;
        else if (c == 2) l64[(uint32_t)i - r0 - r1] = (uint32_t)i;           // clusters in front of i that are of neither of the first two classes
        __syncthreads();
        if (threadIdx.x == 0) { s_c0 += t0; s_c1 += t1; }
        __syncthreads();
    }
}
// SUB = lanes per cluster (32: two clusters per wave, 16: four).  `list` != nullptr: the clusters a narrower instantiation flagged.
#ifdef PS_STOP                        // cumulative cost of the phases (tools/pair_stop.sh): the kernel ends at tick PS_STOP
#define PS_TICK(k, live_) do { if ((k) >= PS_STOP) { if ((uint32_t)(live_) == 0xDEADBEEFu) flag_out[0] = 1; return; } } while (0)      // (live_: what the phase computed -- keeps it from being optimised away)
#else
#define PS_TICK(k, live_) do { } while (0)
#endif
template <int SUB>
__global__ __launch_bounds__(256) void k_pairing_sub(DevBatch b, DevParams p, Work w, uint32_t n_clusters, const uint32_t *list, const unsigned long long *list_n, uint8_t *flag_out, int direct) {
    constexpr int PER = 64 / SUB;
    const int lane = lane_id(), hl = lane & (SUB - 1), hb = lane & ~(SUB - 1);
    const uint32_t idx = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * PER + (uint32_t)(lane / SUB);
    const uint32_t total = list ? (uint32_t)*list_n : n_clusters;
    if ((blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * PER >= total) return;
    bool live = idx < total;
    const uint32_t c = live ? (list ? list[idx] : idx) : 0u;
    uint32_t start = 0, n = 0; int thr = 0;
    if (live) {
        start = w.cl_start[c]; n = w.cl_n[c];
        const uint32_t mode = d_thr_mode(w.cl_ikey[c], w.si, p);
        if (mode == THR_NEVER) { if (hl == 0) { w.cl_npairs[c] = 0; w.cl_ngroups[c] = 0; w.cl_hasumi[c] = 0; } live = false; }   // gencore.cpp:23
        thr = mode == THR_PROPER ? p.proper_thr : p.unproper_thr;
    }
    uint32_t my = NONE32; int nl = 0; const char *nm = nullptr; int ul = 0;
    const char *up = nullptr;
    if (live && n <= (uint32_t)SUB && hl < (int)n) {
        my = w.members[start + hl];
        const uint64_t ui_ = w.uinfo[my];                          // name length, UMI place and length in one word (round 5: core record, UMI pointer and UMI length were three more sectors per read)
        nl = (int)((ui_ >> 41) & 0xFFu) - 1;
        nm = d_qname(b, my);
        ul = uinfo_len(ui_);
        up = uinfo_ptr(b, ui_);
    }
    const uint32_t toolong = sub_ballot<SUB>(nl > 64 || ul > 24, hb);
    if (live && (n > (uint32_t)SUB || toolong)) {                                                  // the next wider kernel takes it -- or, when the tiers were given their clusters by size
        if (hl == 0) { if (direct) w.left_list[atomicAdd(&w.si->n_slow_pair2, 1u)] = c; else flag_out[c] = 1; }     // beforehand (direct: they run side by side), the generic kernels: what is left here then has
        live = false;                                                                              // names beyond 64 bytes or UMIs beyond 24, which every tier up to those hands on
    }
    if (!__any(live)) return;
    PS_TICK(0, my ^ (uint32_t)nl ^ (uint32_t)ul ^ (uint32_t)(uintptr_t)nm ^ (uint32_t)(uintptr_t)up);
    const bool act = live && hl < (int)n;
    const int nwords = (wave_max_u(act ? nl : 0) + 7) >> 3;
    uint64_t nw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t v = 0;
        if (k < nwords) {
            const int rem = (act ? nl : 0) - 8 * k;
            if (rem > 0) { v = *(const u64_unaligned *)(nm + 8 * k); if (rem < 8) v &= (1ull << (8 * rem)) - 1ull; v = d_bswap64(v); }
        }
        nw[k] = v;
    }
    uint64_t ruw[3];                                        // the UMI's bytes travel with the name's
    load_be_words<3>(act ? up : nullptr, act ? ul : 0, ruw);
    // ---- name classes: 32-bit hash as a filter (verified word by word below)
    uint32_t h32 = 0x9E3779B9u;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < nwords) {
        const uint32_t lo = (uint32_t)nw[k], hi = (uint32_t)(nw[k] >> 32);
        h32 = (__builtin_rotateleft32(h32, 5) ^ lo) + hi;
        h32 = __builtin_rotateleft32(h32, 11) ^ (hi + 0x7F4A7C15u);
    }
    h32 ^= h32 >> 15;
    PS_TICK(1, h32 ^ (uint32_t)(nw[0] ^ nw[1] ^ nw[2] ^ nw[3] ^ nw[4] ^ nw[5] ^ nw[6] ^ nw[7]) ^ (uint32_t)(ruw[0] ^ ruw[1] ^ ruw[2]));
    const int nmax = wave_max_u(live ? (int)n : 0);
    uint32_t EQ = 0, LOW = 0;
    {   // one ballot per DISTINCT hash of a half: a whole name class at a time (the halves run to the largest class count)
        uint32_t todo = sub_ballot<SUB>(act, hb);
        while (__any(todo != 0)) {
            const int j = todo ? __ffs((int)todo) - 1 : hl;
            const uint32_t oh = (uint32_t)__shfl((int)h32, hb + j);
            const bool same = act && todo != 0 && h32 == oh;
            const uint32_t cls = sub_ballot<SUB>(same, hb);
            if (same) EQ = cls;
            todo &= ~cls;
        }
    }
    EQ &= ~(1u << hl);
    {   // exact verification of every hash match, arrival order inside the class (all lanes run the shuffles)
        bool bad = false;
        const int rounds = wave_max_u(act ? __popc(EQ) : 0);
        uint32_t rest = act ? EQ : 0u;
        for (int r = 0; r < rounds; r++) {
            const bool has = rest != 0;
            const int sl = has ? __ffs((int)rest) - 1 : hl;
            rest &= rest - 1;
            const uint32_t oj = (uint32_t)__shfl((int)my, hb + sl);
            if (has && oj < my) LOW |= 1u << sl;
#pragma unroll
            for (int k = 0; k < 8; k++) if (k < nwords) { const uint64_t o = shfl64(nw[k], hb + sl); if (has && o != nw[k]) bad = true; }
        }
        const uint32_t badm = sub_ballot<SUB>(bad, hb);
        if (live && badm) { if (hl == 0) { if (direct) w.left_list[atomicAdd(&w.si->n_slow_pair2, 1u)] = c; else w.slow_list[atomicAdd(&w.si->n_slow_pair, 1u)] = c; } live = false; }     // false hash match: generic kernel (direct: slow_list is being consumed on the other stream)
    }
    if (!__any(live)) return;
    PS_TICK(2, EQ ^ (LOW << 1));
    const bool act2 = act && live;
    // ---- pairs: first read of a name = mLeft, last one = mRight
    const bool first = act2 && !(EQ & LOW), last = act2 && !(EQ & ~LOW);
    const uint32_t FIRST = sub_ballot<SUB>(first, hb);
    const uint32_t npairs = __popc(FIRST);
    // ---- std::map order: lexicographic compares only against the first read of every OTHER name.
    //      The names of a cluster share a long prefix (instrument, run, lane, often the tile): the 8 bytes behind the cluster's common
    //      prefix, one big-endian word per read, decide nearly every comparison -- one broadcast per other name instead of one per
    //      name WORD.  Names whose order words tie (they differ later, or one is a prefix of the other) take the full compare.
    uint32_t LT = 0;
    {
        int cpb = 64;                                       // bytes my name shares with the half's first read
#pragma unroll
        for (int k = 7; k >= 0; k--) if (k < nwords) { const uint64_t x = shfl64(nw[k], hb) ^ nw[k]; if (x) cpb = 8 * k + (__clzll((long long)x) >> 3); }
        cpb = act2 ? cpb : 64;
        int cp = -sub_max<SUB>(-cpb);                       // common prefix of the half (64: every name equal)
        cp = min(cp, 56);
        uint64_t okey;                                      // name bytes [cp, cp + 8)
        {
            const int wi = cp >> 3, sh = 8 * (cp & 7);
            uint64_t a = 0, c2 = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { if (k == wi) a = nw[k]; if (k == wi + 1) c2 = nw[k]; }
            okey = sh ? ((a << sh) | (c2 >> (64 - sh))) : a;
        }
        const int itmax = wave_max_u((int)npairs);
        uint32_t fm = FIRST, TIE = 0;
        for (int it = 0; it < itmax; it++) {
            const bool has = fm != 0;
            const int j = has ? __ffs((int)fm) - 1 : hl;
            fm &= fm - 1;
            const uint64_t o = shfl64(okey, hb + j);
            if (has && o < okey) LT |= 1u << j;
            if (has && o == okey && !((EQ >> j) & 1u) && j != hl) TIE |= 1u << j;      // another name with my order word
        }
        if (__any(act2 && TIE != 0)) {                      // (rare) settle the ties on the whole names
            const int rounds = wave_max_u(__popc(TIE));
            uint32_t tm = act2 ? TIE : 0u;
            for (int it = 0; it < rounds; it++) {
                const bool has = tm != 0;
                const int j = has ? __ffs((int)tm) - 1 : hl;
                tm &= tm - 1;
                int cmp = 0;                                // sign of name_j - name_mine
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k < nwords) {
                        const uint64_t o = shfl64(nw[k], hb + j);
                        if (cmp == 0) cmp = o < nw[k] ? -1 : (o > nw[k] ? 1 : 0);
                    }
                }
                if (has && cmp < 0) LT |= 1u << j;
            }
        }
    }
    const uint32_t pidx = __popc(LT);                       // distinct names before mine
    PS_TICK(3, LT ^ pidx ^ FIRST);
    {   // setRight (pair.cpp:201-212): the UMI must equal the pair's current UMI if that is non-empty; the pair's current read is
        // the predecessor in arrival order = the largest read index among the same-name reads before mine
        uint32_t prev = act2 ? (EQ & LOW) : 0u;
        const int rounds = wave_max_u(__popc(prev));
        int plane = -1; uint32_t pv = 0;
        for (int r = 0; r < rounds; r++) {
            const bool has = prev != 0;
            const int sl = has ? __ffs((int)prev) - 1 : hl;
            prev &= prev - 1;
            const uint32_t o = (uint32_t)__shfl((int)my, hb + sl);
            if (has && (plane < 0 || o > pv)) { pv = o; plane = sl; }
        }
        if (rounds > 0) {
            const int src = hb + (plane < 0 ? hl : plane);
            const uint64_t q0 = shfl64(ruw[0], src);
            const int qul = __shfl(ul, src);
            bool same = qul == ul && q0 == ruw[0];
            if (wave_max_u(act2 ? ul : 0) > 8) { const uint64_t q1 = shfl64(ruw[1], src), q2 = shfl64(ruw[2], src); same = same && q1 == ruw[1] && q2 == ruw[2]; }     // (wave-uniform)
            if (plane >= 0 && qul != 0 && !same) raise_error(w.si, GCE_ERR_UMI_MISMATCH, my);
        }
    }
    const bool any_umi = sub_ballot<SUB>(act2 && last && ul > 0, hb) != 0;
    // ---- lanes now stand for pairs (qname order): every read pushes its fields to its pair's lane.  Lanes with nothing to send aim
    //      at lane 31 of the half, a pair lane only when all 32 reads are mate-less singletons -- and then every lane sends.
    const bool pact = live && hl < (int)npairs;
    uint32_t L, R, g_of = 0, ngroups = live ? 1u : 0u;
    {
        const int to_first = (hb + (first ? (int)pidx : SUB - 1)) << 2, to_right = (hb + ((last && !first) ? (int)pidx : SUB - 1)) << 2;
        L = (uint32_t)__builtin_amdgcn_ds_permute(to_first, first ? (int)(my + 1u) : 0) - 1u;
        R = (uint32_t)__builtin_amdgcn_ds_permute(to_right, (last && !first) ? (int)(my + 1u) : 0) - 1u;
        if (!pact) { L = NONE32; R = NONE32; }
    }
    PS_TICK(4, L ^ (R << 1) ^ (uint32_t)any_umi);
    if (__any(any_umi)) {                                    // greedy UMI grouping (cluster.cpp:57-100), for the halves that carry UMIs
        uint64_t uw[3]; int ulen;
        {
            const int to_last = (hb + (last ? (int)pidx : SUB - 1)) << 2;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute(to_last, last ? (int)(uint32_t)ruw[k] : 0);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_permute(to_last, last ? (int)(uint32_t)(ruw[k] >> 32) : 0);
                uw[k] = pact ? (((uint64_t)hi << 32) | lo) : 0ull;
            }
            ulen = __builtin_amdgcn_ds_permute(to_last, last ? ul : 0);
            if (!pact) ulen = 0;
        }
        const bool grp = any_umi;                           // (a half without UMIs keeps its single group)
        int cnt = 0, urank = 0;                             // umiCount[umi], and the rank of my UMI in std::string order
        const bool one_word = wave_max_u(ulen) <= 8;
        {   // one round per DISTINCT UMI of a half: its pairs learn their count, the pairs with a larger UMI add it to their rank
            uint32_t todo = sub_ballot<SUB>(pact, hb);
            if (!grp) todo = 0u;
            while (__any(todo != 0)) {
                const bool open = todo != 0;
                const int q = open ? __ffs((int)todo) - 1 : hl;
                const uint64_t a0 = shfl64(uw[0], hb + q);
                bool eq, lt;
                // (UMI characters are [ATCG_], never NUL: zero-padded big-endian words order and separate UMIs of different lengths
                //  by themselves -- "AC" < "ACG", and equal words mean equal lengths; no length broadcast in the usual one-word case)
                if (one_word) { eq = a0 == uw[0]; lt = a0 < uw[0]; }
                else {
                    const int al = __shfl(ulen, hb + q);
                    const uint64_t a1 = shfl64(uw[1], hb + q), a2 = shfl64(uw[2], hb + q);
                    eq = a0 == uw[0] && a1 == uw[1] && a2 == uw[2] && al == ulen;
                    lt = a0 != uw[0] ? a0 < uw[0] : (a1 != uw[1] ? a1 < uw[1] : (a2 != uw[2] ? a2 < uw[2] : al < ulen));
                }
                const uint32_t cls = sub_ballot<SUB>(open && pact && eq, hb);
                const int sz = __popc(cls);
                if (open && pact && eq) cnt = sz;
                if (open && pact && lt) urank += sz;
                todo &= ~cls;
            }
        }
        if (grp) { g_of = NONE32; ngroups = 0; }
        uint32_t remaining = grp ? sub_ballot<SUB>(pact, hb) : 0u;
        while (__any(remaining != 0)) {
            const bool open = remaining != 0;
            const int key = (open && pact && g_of == NONE32) ? (cnt * 64 + (63 - urank)) : -1;     // highest count, then smallest UMI
            const int best = sub_max<SUB>(key);
            const uint32_t bm = sub_ballot<SUB>(key == best && key >= 0, hb);
            const int tl = hb + (bm ? __ffs((int)bm) - 1 : hl);
            const uint64_t t0 = shfl64(uw[0], tl);
            int diff = popc_nonzero_bytes(t0 ^ uw[0]);                                           // Cluster::umiDiff
            if (!one_word) { const uint64_t t1 = shfl64(uw[1], tl), t2 = shfl64(uw[2], tl); diff += popc_nonzero_bytes(t1 ^ uw[1]) + popc_nonzero_bytes(t2 ^ uw[2]); }   // (wave-uniform: UMIs of <= 8 bytes end in the first word)
            const bool take = open && pact && g_of == NONE32 && diff <= thr;
            if (take) g_of = ngroups;
            remaining &= ~sub_ballot<SUB>(take, hb);
            if (open) ngroups++;
        }
    }
    PS_TICK(5, g_of ^ (ngroups << 8) ^ L ^ R);
    // ---- lay the pairs out group by group (qname order inside a group)
    {
        uint32_t gbase = 0;
        const int gmax = wave_max_u((int)ngroups);
        for (int g = 0; g < gmax; g++) {
            const bool in = pact && g_of == (uint32_t)g;
            const uint32_t m = sub_ballot<SUB>(in, hb);
            if (in) { const uint32_t d = start + gbase + __popc(m & ((1u << hl) - 1u)); w.gpl[d] = L; w.gpr[d] = R; }
            const uint32_t run = __popc(m);
            if (live && hl == 0 && g < (int)ngroups) { w.grp_begin[start + g] = start + gbase; w.grp_n[start + g] = run; }
            gbase += run;
        }
    }
    if (live && hl == 0) {
        const bool cross = d_key(b.core[w.members[start]], p).right < 0;
        w.cl_npairs[c] = npairs; w.cl_ngroups[c] = ngroups; w.cl_hasumi[c] = (uint8_t)((any_umi ? 1 : 0) | (cross ? 2 : 0));
    }
}
