// gce_lean2.hpp — both sides of one group in ONE wave: lanes 0-31 vote the left reads, lanes 32-63 the right reads.
//
// The per-side consensus (consensus_fast_side<true>) is bound by the serial latency of one wave's own instruction stream:
// a side keeps 6 lanes busy while it gathers its reads, 38 while it votes, ~9 while it decides.  Running the two sides of a
// group as the two halves of a wave executes that stream once for both.  Everything that was wave-uniform per side (template,
// voter list, masks, loop bounds) becomes half-uniform: kept in VGPRs, broadcast inside a half with ds_bpermute, and every
// loop runs to the larger of the two halves' trip counts under per-lane predicates.
//
// Scope: <= 32 pairs; the side's reads are one class with the same single-M CIGAR and length, plus at most a minority of reads
// that are provably unrelated to it (a soft clip or indel: see the classification below) -- or one class with the same 2-/3-op
// CIGAR and nothing else; right side: equal positions, i.e. leftReadMode; packed-byte vote applicable.  A side that does not qualify is
// flagged for the full per-side kernel; the other half carries on.
//
// One lane = 8 consecutive columns = 4 packed-base bytes + 8 quals + 8 scores (32 lanes x 8 = 256 columns).
#pragma once

__device__ __forceinline__ uint32_t half_ballot(bool pr, int h) {
    const unsigned long long m = __ballot(pr);
    return h ? (uint32_t)(m >> 32) : (uint32_t)m;
}
__device__ __forceinline__ int half_sum(int v) {            // xor offsets < 32 stay inside the half
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// packed byte max of two words whose bytes are < 128 on the `b` side after masking (see pass A)
__device__ __forceinline__ uint32_t bytemax4(uint32_t a, uint32_t bq) {
    const uint32_t ge = ((a | 0x80808080u) - bq) & 0x80808080u;          // per byte: a >= b
    const uint32_t sel = (ge - (ge >> 7)) | ge;                           // 0xFF where a >= b
    return (a & sel) | (bq & ~sel);
}
// bytes [a_, z_) of a 4-byte unit as a mask
__device__ __forceinline__ uint32_t byte_range4(int a_, int z_) {
    a_ = max(a_, 0); z_ = min(z_, 4);
    if (z_ <= a_) return 0u;
    return (z_ >= 4 ? 0xFFFFFFFFu : ((1u << (8 * z_)) - 1u)) & ~((1u << (8 * a_)) - 1u);
}

// the four-column accept test of pass A on packed bytes (same arithmetic as consensus_fast_side): returns the contested flags
// (bit 7 of byte k = column k of the unit) and the unit's template nibbles, one per byte
__device__ __forceinline__ uint32_t accept4(const DevParams &p, uint32_t t16, uint32_t dacc16, uint32_t ssum, uint32_t tqm, uint32_t cnt4,
                                            int accept_score, uint32_t bmask, uint32_t &tn4) {
    tn4 = __builtin_amdgcn_perm(0u, ((t16 >> 4) & 0x0F0Fu) | ((t16 & 0x0F0Fu) << 16), 0x03010200u);
    const uint32_t dn4 = __builtin_amdgcn_perm(0u, ((dacc16 >> 4) & 0x0F0Fu) | ((dacc16 & 0x0F0Fu) << 16), 0x03010200u);
    const uint32_t differ = (dn4 + 0x7F7F7F7Fu) & 0x80808080u;
    const uint32_t sel = tn4 & 0x07070707u;
    const uint32_t v_lo = __builtin_amdgcn_perm(0x000000FFu, 0x00FFFF00u, sel);      // nibbles 1,2,4
    const uint32_t v_hi = __builtin_amdgcn_perm(0xFF000000u, 0x000000FFu, sel);      // nibbles 8,15
    const uint32_t hi8 = ((tn4 >> 3) & 0x01010101u) * 0xFFu;
    const uint32_t valid = (v_hi & hi8) | (v_lo & ~hi8);
    const uint32_t ge_q = ((tqm | 0x80808080u) - 0x01010101u * (uint32_t)p.moderate_q) & 0x80808080u;
    const uint32_t rhs = cnt4 * (uint32_t)p.score_bias + 0x01010101u * (uint32_t)accept_score;
    const uint32_t ge_e = (((ssum & 0x00FF00FFu) | 0x01000100u) - (rhs & 0x00FF00FFu)) & 0x01000100u;
    const uint32_t ge_o = ((((ssum >> 8) & 0x00FF00FFu) | 0x01000100u) - ((rhs >> 8) & 0x00FF00FFu)) & 0x01000100u;
    const uint32_t ge_s = (ge_e >> 1) | (ge_o << 7);
    return ~(ge_q & ge_s & ~differ & valid) & bmask & 0x80808080u;
}

#define L2_HALF_BYTES 2400      // per half: new base [256], new qual [256], contested columns u16[256], tallies [16][5][4] u32, voter lanes [32], {reference pointer, length, template pos / first CIGAR word / op count}

__device__ void consensus_lean_pair(const DevBatch &b, const DevParams &p, const Work &w, uint32_t gi, uint8_t *s_wave, int lane) {
    const uint32_t begin = w.g_begin[gi], np = w.g_np[gi];
    const int h = lane >> 5, hl = lane & 31, hb = lane & 32;
    if (np == 1 && w.gpr[begin] == NONE32) {                                  // group.cpp:73-77: returned untouched
        if (lane == 0) { w.rp_left[gi] = w.gpl[begin]; w.rp_right[gi] = NONE32; }
        return;
    }
    if (np > 32 || (int)np > p.skip_low_complexity_thr) {                     // deep group: the per-side kernel (and from there the generic one)
        if (lane == 0) { w.gen_flag[gi * 2] = 1; w.gen_flag[gi * 2 + 1] = 1; }
        return;
    }
    uint32_t *rp_out = h ? w.rp_right : w.rp_left;
    const uint32_t *side = h ? w.gpr : w.gpl;
    // ---- per-lane read metadata
    const uint32_t rd = hl < (int)np ? side[begin + hl] : NONE32;
    const bool has = rd != NONE32;
    int pos = 0, lq = 0, nc = 0, isz = 0, tid16 = 0; uint32_t c0 = 0, patch = 0; uint64_t so = 0, qo = 0;
    if (has) {
        patch = w.spatch[rd];
        const ReadDesc k = load_desc(w.rdesc, rd);
        pos = k.pos; lq = k.lq; nc = k.nc; isz = k.isize; so = k.so; qo = k.qo; c0 = k.c0; tid16 = k.tid16;
    }
    const uint32_t hm = half_ballot(has, h);                                  // the reads of my side
    bool done = hm == 0;                                                      // nothing (more) to do for this half
    uint32_t result = NONE32; bool write_result = true;                       // what lane 0 of the half stores into rp_left / rp_right
    // The side's majority class = CIGAR and length of its first single-M read.  group.cpp:177-261 collapse to "containedBy = class
    // size for the class, template = its first read in qname order, voters = the class" when
    //   - every other read is provably unrelated to the class: it has >= 2 CIGAR ops (so it is part of no class read) and its first
    //     op is not an M block of >= len bases (so no class read is part of it; lean sides are always in leftReadMode) -- a soft
    //     clip or an indel among "150M" reads, typically -- and there are fewer of them than class reads (containedBy <= their
    //     number), and
    //   - right side: all positions are equal (leftReadMode, group.cpp:177-194).
    const uint32_t single = half_ballot(has && nc == 1 && cig_op(c0) == 0, h);
    // No single-M read at all: the side may still be one class with a 2- or 3-op CIGAR (all duplicates of a molecule carry its
    // soft clip or indel).  Then every read must belong to the class; the remaining CIGAR words are fetched for the comparison.
    const bool multi = single == 0;
    uint32_t cw1 = 0, cw2 = 0;
    if (has && multi && nc >= 2 && nc <= 3) { const uint32_t *cg = b.cigar + b.cigar_off[rd]; cw1 = cg[1]; if (nc == 3) cw2 = cg[2]; }
    const int fl = hb + (multi ? (hm ? __ffs((int)hm) - 1 : 0) : __ffs((int)single) - 1);             // the template
    const uint32_t o_c0 = (uint32_t)__shfl((int)c0, fl), o_cw1 = (uint32_t)__shfl((int)cw1, fl), o_cw2 = (uint32_t)__shfl((int)cw2, fl);
    const int len = __shfl(lq, fl), o_pos = __shfl(pos, fl), o_nc = __shfl(nc, fl);
    const bool major = has && nc == o_nc && c0 == o_c0 && cw1 == o_cw1 && cw2 == o_cw2 && lq == len;
    const uint32_t vm = half_ballot(major, h);                                // the voters
    const bool unfit = has && ((!major && (multi || nc < 2 || (cig_op(c0) == 0 && cig_len(c0) >= len))) || (h == 1 && pos != o_pos) || o_nc > 3 || o_nc < 1);
    const int nvot = __popc(vm);
    const int accept_score = max(p.base_score_req, 1);
    const uint32_t unfit_m = half_ballot(unfit, h);
    bool to_gen = !done && (unfit_m != 0 || nvot <= __popc(hm) - nvot || len > 256 || !p.q2s_swar_ok ||
                            nvot * (p.score_max + p.score_bias) > 255 || accept_score + nvot * p.score_bias > 255);
    if (to_gen) {
        if (hl == 0) w.gen_flag[gi * 2 + h] = 1;                               // (a flag, compacted afterwards: one shared counter would serialise)
        done = true; write_result = false;
    }
    if (!done && (double)nvot < (double)np * 0.4 && np != 1) done = true;     // group.cpp:264-266: result stays NONE
    const uint32_t out = (uint32_t)__shfl((int)rd, fl);
    const uint64_t o_so = (uint64_t)__shfl((long long)so, fl), o_qo = (uint64_t)__shfl((long long)qo, fl);
    const int o_isz = __shfl(isz, fl), o_t16 = __shfl(tid16, fl);
    if (!__any(!done)) {                                                      // both halves settled
        if (hl == 0 && write_result) rp_out[gi] = result;
        return;
    }
    const int nbytes = (len + 1) >> 1;
    uint8_t *sh = s_wave + h * L2_HALF_BYTES;
    unsigned long long *refslot = (unsigned long long *)(sh + 2336);          // {reference pointer, length}: parked in LDS until pass B
    const uint8_t *ref = nullptr; int64_t ref_len = 0;
    if (!done) {
        const int o_tid = o_t16 != 0xFFFF ? o_t16 : b.core[out].tid;
        if (o_isz != 0 && o_tid >= 0 && o_tid < p.n_ref) {                    // group.cpp:362-367 -> Reference::getData
            const uint8_t *rdp = p.ref_data[o_tid];
            const int64_t need_len = (int64_t)(o_nc == 1 ? ((len - 1) < cig_len(o_c0) ? (len - 1) : -1) : d_ref_offset(b.cigar + b.cigar_off[out], o_nc, len - 1)) + 1;
            if (rdp && (int64_t)o_pos + need_len < p.ref_len[o_tid]) { ref = rdp; ref_len = p.ref_len[o_tid]; }
        }
    }
    if (hl == 0) { refslot[0] = (unsigned long long)ref; refslot[1] = (unsigned long long)ref_len; ((int *)refslot)[4] = o_pos; ((uint32_t *)refslot)[5] = o_c0; ((int *)refslot)[6] = o_nc; }
    if (hl == 0) { refslot[4] = o_so; refslot[5] = o_qo; ((uint32_t *)refslot)[7] = out; }       // template offsets + read: reloaded for the write-back
    uint8_t *resb = sh, *resq = sh + 256;
    uint16_t *cplx = (uint16_t *)(sh + 512);
    uint32_t *tl = (uint32_t *)(sh + 1024);                                   // [16 columns][5 bins][cnt, score, qualsum, topqual]
    uint8_t *vlist = sh + 1024 + 16 * 5 * 16;                                 // voter lanes (absolute) in ascending order
    if (major) vlist[__popc(vm & ((1u << hl) - 1u))] = (uint8_t)lane;
    const int nv_max = max(__builtin_amdgcn_readlane(done ? 0 : nvot, 0), __builtin_amdgcn_readlane(done ? 0 : nvot, 32));
    WAVE_SYNC();
    // ---- pass A: every column.  Early accept (group.cpp:421-428) for a column whose voters all show the template's A/C/G/T/N
    //      nibble with score sum >= baseScoreReq and top quality >= moderate; everything else is queued for pass B.
    const int c8 = 8 * hl;
    const bool act = !done && c8 < len;
    const int nval = act ? min(8, len - c8) : 0;
    const uint32_t nmask = nval >= 8 ? 0xFFFFFFFFu : (((1u << (8 * (nval >> 1))) - 1u) | ((nval & 1) ? (0xF0u << (8 * (nval >> 1))) : 0u));
    const uint32_t bm_lo = byte_range4(0, nval), bm_hi = byte_range4(0, nval - 4);
    uint32_t t32 = 0;
    if (act) t32 = *(const u32_unaligned *)(b.seq + o_so + 4 * hl);
    uint32_t dacc = 0, ss_lo = 0, ss_hi = 0, tq_lo = 0, tq_hi = 0, qor = 0;
    const int s_min = min(min(p.s_high, p.s_moderate), min(p.s_low, p.s_bad));
    const bool lower_bound_ok = nvot * s_min >= accept_score;                 // heuristic only: any lower bound keeps the result exact
    const uint32_t smin4 = 0x01010101u * (uint32_t)((s_min + p.score_bias) & 0xFF);
    const uint32_t smod4 = 0x01010101u * (uint32_t)((p.s_moderate + p.score_bias) & 0xFF);
    for (int k = 0; k < nv_max; k++) {
        const int vl = vlist[k < nvot ? k : 0];                               // (all lanes run the shuffles)
        const uint64_t vso = (uint64_t)__shfl((long long)so, vl), vqo = (uint64_t)__shfl((long long)qo, vl);
        const uint32_t vpatch = (uint32_t)__shfl((int)patch, vl);
        if (act && k < nvot) {
            const uint32_t s32 = *(const u32_unaligned *)(b.seq + vso + 4 * hl);
            const uint32_t q_lo = *(const u32_unaligned *)(b.qual + vqo + c8), q_hi = *(const u32_unaligned *)(b.qual + vqo + c8 + 4);
            // scores: qual2score (or its lower bound, see consensus_fast_side) outside the voter's mate-overlap patch, the stored
            // bytes inside it, a constant for a read scored without a usable mate
            uint32_t sc_lo, sc_hi;
            if (vpatch == GCE_PATCH_CONST) { sc_lo = smod4; sc_hi = smod4; }
            else {
                sc_lo = lower_bound_ok ? smin4 : d_q2s4_biased(p, q_lo); sc_hi = lower_bound_ok ? smin4 : d_q2s4_biased(p, q_hi);
                const int ps = (int)(vpatch & 0xFFFF) - c8, pe = ps + (int)(vpatch >> 16);       // patch = unit bytes [ps, pe)
                // bytes [ps, pe) of the 8-byte unit as one 64-bit mask (two shifts) instead of two clamped 32-bit ranges
                const int a8 = min(max(ps, 0), 8), z8 = min(max(pe, 0), 8);
                const uint64_t upto_z = z8 >= 8 ? ~0ull : ((1ull << (8 * z8)) - 1ull), upto_a = a8 >= 8 ? ~0ull : ((1ull << (8 * a8)) - 1ull);
                const uint64_t pm8 = z8 > a8 ? (upto_z & ~upto_a) : 0ull;
                const uint32_t pm_lo = (uint32_t)pm8, pm_hi = (uint32_t)(pm8 >> 32);
                if (pm_lo | pm_hi) {
                    const uint32_t st_lo = *(const u32_unaligned *)((const uint8_t *)w.score + vqo + c8), st_hi = *(const u32_unaligned *)((const uint8_t *)w.score + vqo + c8 + 4);
                    sc_lo = (sc_lo & ~pm_lo) | (st_lo & pm_lo); sc_hi = (sc_hi & ~pm_hi) | (st_hi & pm_hi);
                }
            }
            const uint32_t ql = q_lo & bm_lo, qh = q_hi & bm_hi;
            dacc |= (s32 ^ t32) & nmask;
            ss_lo += sc_lo & bm_lo; ss_hi += sc_hi & bm_hi;
            qor |= ql | qh;
            tq_lo = bytemax4(tq_lo, ql); tq_hi = bytemax4(tq_hi, qh);
        }
    }
    bool odd = (qor & 0x80808080u) != 0;
    int n_cplx = 0;
    {
        const uint32_t cnt4 = 0x01010101u * (uint32_t)nvot;                   // every voter covers every column of a uniform side
        uint32_t tn_lo, tn_hi;
        const uint32_t cq_lo = accept4(p, t32 & 0xFFFFu, dacc & 0xFFFFu, ss_lo, tq_lo, cnt4, accept_score, bm_lo, tn_lo);
        const uint32_t cq_hi = accept4(p, t32 >> 16, dacc >> 16, ss_hi, tq_hi, cnt4, accept_score, bm_hi, tn_hi);
        if (act) { *(uint32_t *)(resb + c8) = tn_lo; *(uint32_t *)(resb + c8 + 4) = tn_hi; *(uint32_t *)(resq + c8) = tq_lo; *(uint32_t *)(resq + c8 + 4) = tq_hi; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool cq = act && (((k < 4 ? cq_lo : cq_hi) >> (8 * (k & 3) + 7)) & 1u);
            const uint32_t mk = half_ballot(cq, h);
            if (cq) cplx[n_cplx + __popc(mk & ((1u << hl) - 1u))] = (uint16_t)(c8 + k);
            n_cplx += __popc(mk);
        }
    }
    WAVE_SYNC();
    // ---- pass B: the contested columns, 16 per half and round.  Work items = (column, voter): one lane fetches one voter's
    //      (base, qual, score) for one column and adds it to that column's 5-bin tally (LDS atomics); then one lane per column runs
    //      the rule cascade + reference arbitration.
    int minc = 0;
    {
        const uint32_t magic = ((1u << 20) + (uint32_t)max(nvot, 1) - 1) / (uint32_t)max(nvot, 1);   // item / nvot == (item * magic) >> 20 for item < 2^11
        const int ncplx_mine = done ? 0 : n_cplx;
        const int rounds = (max(__builtin_amdgcn_readlane(ncplx_mine, 0), __builtin_amdgcn_readlane(ncplx_mine, 32)) + 15) >> 4;
        for (int rnd = 0; rnd < rounds; rnd++) {
            const int cbase = 16 * rnd;
            const int ncol = min(16, max(ncplx_mine - cbase, 0));
            for (int k = hl; k < 16 * 5; k += 32) *(uint4 *)(tl + 4 * k) = make_uint4(0, 0, 0, 0);
            WAVE_SYNC();
            int ref4 = 0;                                                     // requested before the voters' bytes: both in flight together
            const uint8_t *ref = (const uint8_t *)refslot[0]; const int64_t ref_len = (int64_t)refslot[1];
            const int o_pos = ((const int *)refslot)[4], o_nc = ((const int *)refslot)[6]; const uint32_t o_c0 = ((const uint32_t *)refslot)[5];
            if (ref && hl < ncol) {
                const int col = cplx[cbase + hl];
                // (a 2-/3-op class walks its CIGAR from memory: rare, and it keeps two registers free for everyone else)
                const int ro = o_nc == 1 ? (col < cig_len(o_c0) ? col : -1) : d_ref_offset(b.cigar + b.cigar_off[((const uint32_t *)refslot)[7]], o_nc, col);
                if (ro >= 0 && (int64_t)o_pos + ro < ref_len) ref4 = d_ref_nib(ref, (int64_t)o_pos + ro);
            }
            const int items = ncol * nvot;
            const int it_max = max(__builtin_amdgcn_readlane(items, 0), __builtin_amdgcn_readlane(items, 32));
            for (int ibase = 0; ibase < it_max; ibase += 32) {
                const int item = ibase + hl;
                const bool live = item < items;
                const int it_ = live ? item : 0;
                const int c = (int)(((uint32_t)it_ * magic) >> 20), kx = it_ - c * nvot;
                const int vl = vlist[live ? kx : 0], col = cplx[cbase + (live ? c : 0)];
                const uint64_t vso = (uint64_t)__shfl((long long)so, vl), vqo = (uint64_t)__shfl((long long)qo, vl);
                const uint32_t vpatch = (uint32_t)__shfl((int)patch, vl);
                if (live) {
                    const int nb = d_nib(b.seq + vso, col), q = b.qual[vqo + col], sc = d_score_at(p, w.score + vqo, vpatch, col, q);
                    const int bin = nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 8 ? 3 : nb == 15 ? 4 : -1;
                    if (bin < 0 || (q & 0x80)) odd = true;
                    else {
                        uint32_t *t4 = tl + (c * 5 + bin) * 4;
                        atomicAdd(t4, 1u); atomicAdd(t4 + 1, (uint32_t)sc); atomicAdd(t4 + 2, (uint32_t)q); atomicMax(t4 + 3, (uint32_t)q);
                    }
                }
            }
            WAVE_SYNC();
            if (hl < ncol) {
                const int col = cplx[cbase + hl];
                Tally5 t; t.total = 0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint4 v4 = *(const uint4 *)(tl + (hl * 5 + k) * 4);
                    t.cnt[k] = (int)v4.x; t.ss[k] = (int)v4.y; t.qs[k] = (int)v4.z; t.tq[k] = (int)v4.w; t.total += (int)v4.y;
                }
                const ColOut r = decide_column_packed(t, p, resb[col], ref4);
                resb[col] = (uint8_t)r.base; resq[col] = (uint8_t)r.qual; minc += r.minc;
            }
            WAVE_SYNC();
        }
    }
    WAVE_SYNC();
    const uint32_t oddm = half_ballot(odd, h);
    if (!done && oddm != 0) {                                  // IUPAC nibble or qual >= 128 among the voters: generic kernel
        if (hl == 0) w.slow_list[atomicAdd(&w.si->n_slow, 1u)] = gi * 2 + h;
        done = true; write_result = false;
    }
    minc = half_sum(done ? 0 : minc);
    if (!done) {
        const uint32_t out = ((const uint32_t *)refslot)[7];
        uint8_t *oseq = b.seq + refslot[4], *oqual = b.qual + refslot[5];
        bool restore = false;
        if (minc != 0) {                                                      // group.cpp:528-573
            const int o_nm_type = b.nm_type[out], o_nm = b.nm[out];
            if (o_nm_type == 0) { if (hl == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
            else if (minc > 5) restore = true;
            else if (hl == 0) { const int nn = o_nm + minc; if (o_nm_type == 'C' && nn >= 0 && nn <= 255) w.rp_nm[gi * 2 + h] = nn; }
        }
        if (!restore) {
            for (int bi = hl; bi < nbytes; bi += 32) {
                const int c = 2 * bi;
                if (c + 1 < len) { oseq[bi] = (uint8_t)((resb[c] << 4) | resb[c + 1]); *(u16_unaligned *)(oqual + c) = (uint16_t)(resq[c] | (resq[c + 1] << 8)); }
                else { oseq[bi] = (uint8_t)((resb[c] << 4) | (oseq[bi] & 0xF)); oqual[c] = resq[c]; }
            }
        }
        result = out;
    }
    if (hl == 0 && write_result) rp_out[gi] = result;
}

// one wave per group (both sides), every group in order
__global__ __launch_bounds__(256, 6) void k_consensus_lean2(DevBatch b, DevParams p, Work w, uint32_t n_groups) {
    __shared__ __attribute__((aligned(16))) uint8_t s_res[WAVES_PER_BLOCK][2 * L2_HALF_BYTES];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t idx = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (idx < n_groups) consensus_lean_pair(b, p, w, idx, s_res[wv], lane);
}
