p='/root/repo/gencore_amd/csrc/gce_vote.hpp'
s=open(p).read()
anchor="__device__ __forceinline__ int vb_find(const uint16_t *pre, int n, int it) {"
plan='''// The pairs of every batch in the order k_vote's lanes take them: row `batch` holds VB_MAXP (left read, right read) entries, entry q = the pair at weight position
// VB_W x batch + q (a group's pairs lie at its weight prefix g_wbase onwards; the positions between a small group's last pair and its weight, those in front of the batch's first group
// and behind its last one hold NONE32 / NONE32).  With it the pair lane of k_vote needs nothing of P0 to ask for its descriptors: batch start -> group arrays (P0) and
// plan row -> descriptors (P1) are two chains of two trips side by side where they were one chain of four (vb_start -> g_begin -> gpl / gpr -> rdesc).
// Eight lanes per group: every entry of a row is written by exactly one group (the first group of a batch clears the row's head, the last one its tail).
#ifndef VB_PLAN
#define VB_PLAN 1
#endif
__global__ __launch_bounds__(256) void k_vote_plan(Work w, const unsigned long long *n_ptr) {
    const uint64_t n = *n_ptr;
    const uint32_t sub = threadIdx.x & 7u;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 3; i < n; i += ((uint64_t)gridDim.x * 256) >> 3) {
        const uint32_t ex = w.g_wbase[i], np = w.g_np[i], gb = w.g_begin[i], wt = (uint32_t)w.gw[i];
        const uint32_t bi = ex / VB_W, local = ex - bi * VB_W;
        const bool first = i == 0 || (ex - (uint32_t)w.gw[i - 1]) / VB_W != bi;          // (k_vote_batches' rule)
        const bool last = i + 1 == n || (ex + wt) / VB_W != bi;
        const uint32_t npl = np > 32u ? 0u : np;                                          // a deep group's pairs are not k_vote's
        uint2 *row = w.vplan + (uint64_t)bi * VB_MAXP;
        const uint32_t lo = first ? 0u : local, hi = last ? (uint32_t)VB_MAXP : local + wt;
        for (uint32_t q = lo + sub; q < hi; q += 8u) {
            uint2 v = make_uint2(NONE32, NONE32);
            if (q >= local && q - local < npl) { v.x = w.gpl[gb + (q - local)]; v.y = w.gpr[gb + (q - local)]; }
            row[q] = v;
        }
    }
}

'''
assert anchor in s
s=s.replace(anchor, plan+anchor,1)
old="""    const uint32_t g0 = w.vb_start[bid_];
    if (g0 == NONE32) return;
"""
new="""    const uint32_t g0 = w.vb_start[bid_];
#if VB_PLAN
    // the batch's pairs (k_vote_plan's row): asked for beside the batch start, so that the descriptors are on their way while P0 looks at the groups
    uint2 lr_ = make_uint2(NONE32, NONE32);
    if (tid < VB_MAXP) lr_ = w.vplan[(uint64_t)bid_ * VB_MAXP + (uint32_t)tid];
#endif
    if (g0 == NONE32) return;
"""
assert old in s; s=s.replace(old,new,1)
old="""    if (tid < 64) {
        const uint32_t gi = g0 + (uint32_t)lane;
        const bool maybe = lane < VB_MAXG && gi < n_groups;                            // (the three loads side by side: whether the group belongs to the batch only decides who uses them)
        const uint32_t wb_ = maybe ? w.g_wbase[gi] : 0u, np_ = maybe ? w.g_np[gi] : 0u, gb_ = maybe ? w.g_begin[gi] : 0u;
        bool in = maybe && wb_ < (bid_ + 1u) * VB_W;
"""
new="""#if VB_PLAN
    uint32_t wb_ = 0u, np_ = 0u, gb_ = 0u;
    {
        const uint32_t gi = g0 + (uint32_t)lane;
        if (tid < 64 && lane < VB_MAXG && gi < n_groups) { wb_ = w.g_wbase[gi]; np_ = w.g_np[gi]; gb_ = w.g_begin[gi]; }
    }
    uint4 dl0_ = make_uint4(0, 0, 0, 0), dl1_ = dl0_, dr0_ = dl0_, dr1_ = dl0_;         // the two descriptors as they come from memory: unpacked behind the barrier
    if (lr_.x != NONE32) { const uint4 *src = reinterpret_cast<const uint4 *>(w.rdesc + lr_.x); dl0_ = src[0]; dl1_ = src[1]; }
    if (lr_.y != NONE32) { const uint4 *src = reinterpret_cast<const uint4 *>(w.rdesc + lr_.y); dr0_ = src[0]; dr1_ = src[1]; }
#endif
    if (tid < 64) {
        const uint32_t gi = g0 + (uint32_t)lane;
        const bool maybe = lane < VB_MAXG && gi < n_groups;                            // (the three loads side by side: whether the group belongs to the batch only decides who uses them)
#if !VB_PLAN
        const uint32_t wb_ = maybe ? w.g_wbase[gi] : 0u, np_ = maybe ? w.g_np[gi] : 0u, gb_ = maybe ? w.g_begin[gi] : 0u;
#endif
        bool in = maybe && wb_ < (bid_ + 1u) * VB_W;
"""
assert old in s; s=s.replace(old,new,1)
old="""        int x = deep ? 0 : (int)np, pre = x;
        pre = wave_scan_incl(pre);
        if (in) { s_ggi[lane] = gi; s_gbeg[lane] = gb_; s_gnp[lane] = (uint8_t)(deep ? 0u : np); s_glp0[lane] = (uint16_t)(pre - x); s_gflag[lane] = deep ? 1 : 0; }
        if (lane == ng - 1) s_np = pre;
"""
new="""#if VB_PLAN
        // a group's pairs lie at its weight position in the batch (the plan's order): no prefix over the pair counts
        const int x = deep ? 0 : (int)np, lp0_ = (int)(wb_ - bid_ * VB_W);
        if (in) { s_ggi[lane] = gi; s_gbeg[lane] = gb_; s_gnp[lane] = (uint8_t)x; s_glp0[lane] = (uint16_t)lp0_; s_gflag[lane] = deep ? 1 : 0; }
        if (lane == ng - 1) s_np = lp0_ + x;                                           // (positions, holes included: < VB_MAXP)
#else
        int x = deep ? 0 : (int)np, pre = x;
        pre = wave_scan_incl(pre);
        if (in) { s_ggi[lane] = gi; s_gbeg[lane] = gb_; s_gnp[lane] = (uint8_t)(deep ? 0u : np); s_glp0[lane] = (uint16_t)(pre - x); s_gflag[lane] = deep ? 1 : 0; }
        if (lane == ng - 1) s_np = pre;
#endif
"""
assert old in s; s=s.replace(old,new,1)
old="""        const uint32_t slot = s_gbeg[j] + (uint32_t)(tid - (int)s_glp0[j]);
        const uint32_t L = w.gpl[slot], R = w.gpr[slot];
        ReadDesc lk{}, rk{};
        if (L != NONE32) lk = load_desc(w.rdesc, L);
        if (R != NONE32) rk = load_desc(w.rdesc, R);
"""
new="""#if VB_PLAN
        // a position behind its group's last pair (the pad of a small group, a group that is handed on at once) holds no pair of this batch
        const bool mine_ = j >= 0 && tid - (int)s_glp0[j] < (int)s_gnp[j];
        const uint32_t L = mine_ ? lr_.x : NONE32, R = mine_ ? lr_.y : NONE32;
        ReadDesc lk{}, rk{};
        if (L != NONE32) lk = unpack_desc(dl0_, dl1_);
        if (R != NONE32) rk = unpack_desc(dr0_, dr1_);
        if (j < 0) j = 0;
#else
        const uint32_t slot = s_gbeg[j] + (uint32_t)(tid - (int)s_glp0[j]);
        const uint32_t L = w.gpl[slot], R = w.gpr[slot];
        ReadDesc lk{}, rk{};
        if (L != NONE32) lk = load_desc(w.rdesc, L);
        if (R != NONE32) rk = load_desc(w.rdesc, R);
#endif
"""
assert old in s; s=s.replace(old,new,1)
old="const uint32_t kb = 1u << (tid - (int)s_glp0[j]);"
new="const uint32_t kb = 1u << ((tid - (int)s_glp0[j]) & 31);                  // (used by the reads of a pair only: a hole of the plan has none)"
assert old in s; s=s.replace(old,new,1)
old="if ((vm >> k) & 1u) s_vlist[side]"
new="if ((unsigned)k < 32u && ((vm >> k) & 1u)) s_vlist[side]"
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='/root/repo/gencore_amd/csrc/gce_device.hpp'
s=open(p).read()
old="__device__ __forceinline__ ReadDesc load_desc(const ReadDescP *base, uint32_t i) {\n    union { ReadDescP d; uint4 q[2]; } u;\n    const uint4 *src = reinterpret_cast<const uint4 *>(base + i);\n    u.q[0] = src[0]; u.q[1] = src[1];\n"
new="""// (the record as it comes from memory, and its unpacking: a kernel that asks for descriptors in front of a barrier and looks at them behind it keeps the two loads in flight)
__device__ __forceinline__ ReadDesc unpack_desc(const uint4 q0, const uint4 q1) {
    union { ReadDescP d; uint4 q[2]; } u;
    u.q[0] = q0; u.q[1] = q1;
    ReadDesc r;
    r.so = (uint64_t)u.d.so_lo | (uint64_t)(u.d.hi & 0xFFu) << 32; r.qo = (uint64_t)u.d.qo_lo | (uint64_t)((u.d.hi >> 8) & 0xFFu) << 32;
    r.c0 = u.d.c0; r.pos = (int32_t)(u.d.pos_fl & 0x7FFFFFFFu); r.isize = (int32_t)(u.d.pos_fl >> 31);
    r.lq = u.d.lq; r.mo = u.d.mo; r.ml = u.d.ml; r.nc = u.d.nc; r.tid = u.d.tid; r.lastm = (uint16_t)(u.d.hi >> 16);
    r.rlen = u.d.nc == 1 ? (int32_t)(cig_len(u.d.c0) * consumes_ref(cig_op(u.d.c0))) : (u.d.nc == 0 ? 0 : RLEN_WALK);
    return r;
}
"""
assert old in s; s=s.replace(old,new+old,1)
open(p,'w').write(s)

p='/root/repo/gencore_amd/csrc/gce_kernels.hpp'
s=open(p).read()
old="    uint64_t *gw; uint32_t *g_wbase, *vb_start;   // k_vote batching: weight of every group, its exclusive prefix, first group of every batch\n"
assert old in s
s=s.replace(old, old+"    uint2 *vplan;                         // k_vote_plan (gce_vote.hpp): per batch VB_MAXP (left read, right read) entries in weight-position order\n",1)
open(p,'w').write(s)

p='/root/repo/gencore_amd/csrc/engine.hip'
s=open(p).read()
old="gw, g_wbase, vb_start, rp_left,"
assert old in s; s=s.replace(old,"gw, g_wbase, vb_start, vplan, rp_left,",1)
old="&e->g_wbase, &e->vb_start, "
assert old in s; s=s.replace(old,"&e->g_wbase, &e->vb_start, &e->vplan, ",1)
anchor="#ifdef VB_STOP\n        {   // experiment builds"
blk="""#if VB_PLAN
        // the batches' pairs in lane order (gce_vote.hpp, k_vote_plan): sized by the batch count the host has just read
        ENS(vplan, (size_t)(nbatch + 8u) * VB_MAXP * sizeof(uint2));
        w.vplan = e->vplan.as<uint2>();
        hipLaunchKernelGGL(k_vote_plan, dim3(std::min<unsigned>(4096u, cdiv((uint64_t)NG * 8u, 256))), dim3(256), 0, s, w, (const unsigned long long *)&w.si->n_groups);
#endif
"""
assert anchor in s
s=s.replace(anchor, blk+anchor,1)
open(p,'w').write(s)
