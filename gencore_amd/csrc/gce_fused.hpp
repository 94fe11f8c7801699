// gce_fused.hpp — the LDS-resident group kernel: one wave per UMI group.
//
//   HBM -> LDS   every read of the group (both mates) is staged ONCE with coalesced dword loads:
//                per read slot  seq[80] | qual[160] | score[160]   (reads up to 160 bases)
//   LDS          Pair::computeScore for every pair (pair.cpp:88-172): score bytes written, mismatching overlap quals rewritten
//   registers    per side: leftReadMode, containedBy, template pick, voters, lenDiff (group.cpp:136-318)
//   LDS          column vote: pass A (unanimous columns, early accept group.cpp:421-428) + pass B (contested columns: 16-bin
//                rule cascade and reference arbitration group.cpp:394-501), mismatchInc reduction, NM patch / restore
//   LDS -> HBM   only the two TEMPLATE reads are written back (seq nibbles, qual row, nm_new): no other read of a group is
//                ever emitted, so their rescored quals never have to leave the CU.
//
// HBM traffic per group = its reads once (229 B/read at 150 bp) + 2 x 225 B out + the reference nibbles of contested
// columns: the algorithmic bytes of SURVEY.md section 8d.  Groups the kernel does not cover (more than NPMAX pairs, reads longer
// than 160 bases, IUPAC nibbles, quals >= 128, groups above the low-complexity threshold) are appended to fb_list and take
// the global-memory kernels (k_score / k_consensus_fast / k_consensus_slow) instead.
#pragma once
#include "gce_kernels.hpp"

#define F_LMAX 160
#define F_SEQB 80
#define F_QOFF 80
#define F_SOFF 240
#define F_RS 400
#define F_RES 768          // resb[160] resq[160] cplx u16[160] (+pad)

__device__ __forceinline__ int lds_nib(const uint8_t *s, int i) { uint8_t v = s[i >> 1]; return (i & 1) ? (v & 0xF) : (v >> 4); }

__device__ inline void fused_defer(const Work &w, uint32_t gi, uint32_t begin, uint32_t np, int lane) {
    for (uint32_t k = lane; k < np; k += 64) w.slot_flag[begin + k] = 1;          // k_score: these pair slots need global scores
    if (lane == 0) w.fb_list[atomicAdd(&w.si->n_fb, 1u)] = gi;
}

template <int NPMAX, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_group_fused(DevBatch b, DevParams p, Work w, uint32_t n_groups, uint32_t np_lo, int last_tier) {
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[WPB][2 * NPMAX * F_RS + F_RES];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t gi = blockIdx.x * WPB + wv;
    if (gi >= n_groups) return;
    const uint32_t begin = w.g_begin[gi], np = w.g_np[gi];
    if (np <= np_lo || (np > (uint32_t)NPMAX && !last_tier)) return;              // another tier's group
    if (np > (uint32_t)NPMAX || (int)np > p.skip_low_complexity_thr) { fused_defer(w, gi, begin, np, lane); return; }
    uint8_t *LD = s_lds[wv];
    uint8_t *resb = LD + 2 * NPMAX * F_RS, *resq = resb + 160;
    uint16_t *cplx = (uint16_t *)(resb + 320);
    // ---- per-lane read metadata: lanes [0,np) = left reads, [np,2np) = right reads (pairs in qname order)
    const int n2 = 2 * (int)np;
    uint32_t rd = NONE32;
    if (lane < n2) rd = lane < (int)np ? w.gpl[begin + lane] : w.gpr[begin + lane - np];
    if (np == 1) {                                                                 // group.cpp:73-77: mate-less singleton returned untouched
        uint32_t r1 = (uint32_t)rl32((int)rd, 1);
        if (r1 == NONE32) { if (lane == 0) { w.rp_left[gi] = rd; w.rp_right[gi] = NONE32; } return; }
    }
    const bool has = rd != NONE32;
    int pos = 0, lq = 0, nc = 0, rrp = 0, mo = 0, ml = 0, isz = 0, tid = -1; uint32_t c0 = 0; uint64_t cigo = 0, so = 0, qo = 0;
    if (has) {
        gce_core k = b.core[rd];
        pos = k.pos; lq = k.l_qseq; nc = k.n_cigar; isz = k.isize; tid = k.tid;
        cigo = b.cigar_off[rd]; so = b.seq_off[rd]; qo = b.qual_off[rd];
        if (nc > 0) c0 = b.cigar[cigo];
        if (nc >= 1 && cig_op(c0) == 0) { mo = 0; ml = cig_len(c0); } else d_first_m(b.cigar + cigo, nc, mo, ml);
        rrp = pos + (nc == 1 ? cig_len(c0) * consumes_ref(cig_op(c0)) : d_cigar_rlen(b.cigar + cigo, nc));
    }
    if (__any(has && lq > F_LMAX)) { fused_defer(w, gi, begin, np, lane); return; }
    const int g_tid = rl32(tid, 0);                                               // every read of a cluster sits on the key's contig
    const uint8_t *ref_base = nullptr; int64_t ref_n = 0;
    if (g_tid >= 0 && g_tid < p.n_ref) { ref_base = p.ref_data[g_tid]; ref_n = p.ref_len[g_tid]; }
    // ---- stage the reads: lanes 0..19 carry seq dwords, lanes 20..59 qual dwords; 8 reads in flight per batch
    bool odd = false;
    for (int s0 = 0; s0 < n2; s0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v[u] = 0;
            const int s = s0 + u;
            if (s < n2) {
                const uint32_t r = (uint32_t)rl32((int)rd, s);
                if (r != NONE32) {
                    const uint64_t so_s = rl64(so, s), qo_s = rl64(qo, s); const int lq_s = rl32(lq, s);
                    const int nsd = (((lq_s + 1) >> 1) + 3) >> 2, nqd = (lq_s + 3) >> 2;
                    if (lane < nsd) v[u] = *(const u32_unaligned *)(b.seq + so_s + 4 * lane);
                    else if (lane >= 20 && lane - 20 < nqd) v[u] = *(const u32_unaligned *)(b.qual + qo_s + 4 * (lane - 20));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int s = s0 + u;
            if (s < n2 && (uint32_t)rl32((int)rd, s) != NONE32) {
                const int lq_s = rl32(lq, s);
                const int nsd = (((lq_s + 1) >> 1) + 3) >> 2, nqd = (lq_s + 3) >> 2;
                uint8_t *slot = LD + s * F_RS;
                if (lane < nsd) {
                    *(uint32_t *)(slot + 4 * lane) = v[u];
#pragma unroll
                    for (int k = 0; k < 8; k++) {                                  // IUPAC nibble among the bases -> generic path
                        const int posn = 8 * lane + k;                             // nibble k of the dword: byte k>>1, high nibble first
                        const int nb = (v[u] >> (8 * (k >> 1) + ((k & 1) ? 0 : 4))) & 0xF;
                        if (posn < lq_s && !((0x8116u >> nb) & 1u)) odd = true;
                    }
                } else if (lane >= 20 && lane - 20 < nqd) {
                    *(uint32_t *)(slot + F_QOFF + 4 * (lane - 20)) = v[u];
#pragma unroll
                    for (int k = 0; k < 4; k++) if (4 * (lane - 20) + k < lq_s && ((v[u] >> (8 * k)) & 0x80)) odd = true;
                }
            }
        }
    }
    if (__any(odd)) { fused_defer(w, gi, begin, np, lane); return; }
    WAVE_SYNC();
    // ---- Pair::computeScore in LDS, one pair after the other, one lane per base
    for (int k = 0; k < (int)np; k++) {
        const int rs_ = (int)np + k;
        const bool pair_has_r = (uint32_t)rl32((int)rd, rs_) != NONE32;
        uint8_t *lslot = LD + k * F_RS, *rslot = LD + rs_ * F_RS;
        const int llen = rl32(lq, k);
        if (!pair_has_r) { for (int i = lane; i < llen; i += 64) lslot[F_SOFF + i] = (uint8_t)p.s_moderate; continue; }
        const int rlen = rl32(lq, rs_);
        const int lmo = rl32(mo, k), lml = rl32(ml, k), rmo = rl32(mo, rs_), rml = rl32(ml, rs_);
        if (!(lml > 0 && rml > 0)) {
            for (int i = lane; i < llen; i += 64) lslot[F_SOFF + i] = (uint8_t)p.s_moderate;
            for (int i = lane; i < rlen; i += 64) rslot[F_SOFF + i] = (uint8_t)p.s_moderate;
            continue;
        }
        const int dis = rl32(pos, rs_) - rl32(pos, k);
        int lstart, rstart, cmp;
        if (dis >= 0) { lstart = lmo + dis; rstart = rmo; cmp = min(lml - dis, rml); }
        else { lstart = lmo; rstart = rmo - dis; cmp = min(lml, rml + dis); }
        for (int l = lane; l < llen; l += 64) {
            const int ql = lslot[F_QOFF + l];
            if (l >= lstart && l < lstart + cmp) {
                const int r = rstart + (l - lstart), qr = rslot[F_QOFF + r];
                if (lds_nib(lslot, l) == lds_nib(rslot, r)) { const int sc = d_qual2score(p, ((ql + qr) / 2) & 0xFF) + 4; lslot[F_SOFF + l] = (uint8_t)sc; rslot[F_SOFF + r] = (uint8_t)sc; }
                else {
                    lslot[F_QOFF + l] = (uint8_t)max(0, ql - qr); rslot[F_QOFF + r] = (uint8_t)max(0, qr - ql);
                    if (ql >= qr) { lslot[F_SOFF + l] = (uint8_t)(d_qual2score(p, ql - qr) - 3); rslot[F_SOFF + r] = 0; }
                    else { lslot[F_SOFF + l] = 0; rslot[F_SOFF + r] = (uint8_t)(d_qual2score(p, qr - ql) - 3); }
                }
            } else lslot[F_SOFF + l] = (uint8_t)d_qual2score(p, ql);
        }
        for (int r = lane; r < rlen; r += 64) if (!(r >= rstart && r < rstart + cmp)) rslot[F_SOFF + r] = (uint8_t)d_qual2score(p, rslot[F_QOFF + r]);
    }
    WAVE_SYNC();
    // ---- both sides: template pick + vote
    const int accept_score = max(p.base_score_req, 1);
#pragma unroll 1
    for (int sidx = 0; sidx < 2; sidx++) {
        const bool is_left = sidx == 0;
        const int sb = is_left ? 0 : (int)np;                                     // lane / slot base of this side
        uint32_t *rp_out = is_left ? w.rp_left : w.rp_right;
        const bool mine = lane >= sb && lane < sb + (int)np;                      // this lane's read belongs to the side
        const bool hs = mine && has;
        const unsigned long long hmask = __ballot(hs);
        if (!hmask) { if (lane == 0) rp_out[gi] = NONE32; continue; }
        bool left_mode = is_left;                                                 // group.cpp:177-194
        if (!is_left) { const int p0 = rl32(pos, __ffsll((long long)hmask) - 1); if (!__any(hs && pos != p0)) left_mode = true; }
        int cb = hs ? 1 : 0;                                                      // containedBy, group.cpp:196-233
        for (unsigned long long m = hmask; m; m &= m - 1) {
            const int j = __ffsll((long long)m) - 1;
            const int wnc = rl32(nc, j), wrrp = rl32(rrp, j); const uint32_t wc0 = (uint32_t)rl32((int)c0, j); const uint64_t wcig = rl64(cigo, j);
            if (hs && lane != j && (is_left || rrp == wrrp) && part_of_fast(c0, nc, b.cigar + cigo, wc0, wnc, b.cigar + wcig, left_mode)) cb++;
        }
        int best = mine ? lane : 0x7FFFFFFF, bc = mine ? cb : -1, bl = hs ? lq : 0;   // group.cpp:235-261 (pair order == lane order)
        for (int o = 32; o > 0; o >>= 1) {
            int ob = __shfl_xor(best, o), oc = __shfl_xor(bc, o), ol = __shfl_xor(bl, o);
            bool better = oc > bc || (oc == bc && (ol < bl || (ol == bl && ob < best)));
            if (better) { best = ob; bc = oc; bl = ol; }
        }
        if ((double)bc < (double)np * 0.4 && np != 1) { if (lane == 0) rp_out[gi] = NONE32; continue; }   // group.cpp:264-266
        const uint32_t out = (uint32_t)rl32((int)rd, best);
        if (out == NONE32) { if (lane == 0) rp_out[gi] = NONE32; continue; }
        const int o_pos = rl32(pos, best), o_lq = rl32(lq, best), o_nc = rl32(nc, best), o_isz = rl32(isz, best); const uint32_t o_c0 = (uint32_t)rl32((int)c0, best);
        const uint64_t o_cigo = rl64(cigo, best), o_so = rl64(so, best), o_qo = rl64(qo, best);
        const uint32_t *ocig = b.cigar + o_cigo;
        bool take = false; int ld = 0;                                            // voters + lenDiff, group.cpp:287-313,339-348
        if (hs) {
            take = lane == best || part_of_fast(o_c0, o_nc, ocig, c0, nc, b.cigar + cigo, left_mode);
            if (take) { ld = lq - o_lq; if (ld != 0 && pos == o_pos && part_of_fast(o_c0, o_nc, ocig, c0, nc, b.cigar + cigo, true)) ld = 0; }
        }
        const unsigned long long vmask = __ballot(take);
        int len = o_lq;
        if (o_nc == 0) len = wave_min(take ? lq : 0x7FFFFFFF);
        const int nbytes = (len + 1) >> 1;
        const uint8_t *ref = nullptr;
        if (o_isz != 0 && ref_base) {                                             // group.cpp:362-367 -> Reference::getData
            int64_t need_len = (int64_t)(o_nc == 1 && cig_op(o_c0) == 0 ? (len - 1 < cig_len(o_c0) ? len - 1 : -1) : d_ref_offset(ocig, o_nc, len - 1)) + 1;
            if ((int64_t)o_pos + need_len < ref_n) ref = ref_base;
        }
        const uint8_t *tslot = LD + best * F_RS;                                  // the template's own slot
        // ---- pass A: all columns (<= 160: three 64-lane slices of single columns would waste lanes; use 2 columns per lane, 2 slices)
        int n_cplx = 0;
        const bool even_ld = left_mode || !__any(take && (ld & 1));
#pragma unroll 1
        for (int it = 0; it * 64 < nbytes; it++) {
            const int bi = it * 64 + lane, col0 = bi * 2;
            const bool a0 = col0 < len, a1 = col0 + 1 < len;
            uint32_t pm0 = 0, pm1 = 0; int ss0 = 0, ss1 = 0, tq0 = 0, tq1 = 0;
            for (unsigned long long m = vmask; m; m &= m - 1) {
                const int v = __ffsll((long long)m) - 1;
                const uint8_t *vs = LD + v * F_RS;
                const int vld = left_mode ? 0 : rl32(ld, v), vlq = rl32(lq, v);
                const int r0 = col0 + vld, r1 = r0 + 1;
                const bool in0 = a0 && r0 >= 0 && r0 < vlq, in1 = a1 && r1 >= 0 && r1 < vlq;
                if (even_ld) {
                    if (in0) {                                                    // r0 even: one seq byte, aligned 16-bit qual / score
                        const uint8_t sbv = vs[r0 >> 1];
                        const uint16_t qq = *(const uint16_t *)(vs + F_QOFF + r0), sc = *(const uint16_t *)(vs + F_SOFF + r0);
                        pm0 |= 1u << (sbv >> 4); ss0 += (int)(int8_t)(sc & 0xFF); tq0 = max(tq0, (int)(qq & 0xFF));
                        if (in1) { pm1 |= 1u << (sbv & 0xF); ss1 += (int)(int8_t)(sc >> 8); tq1 = max(tq1, (int)(qq >> 8)); }
                    }
                } else {
                    if (in0) { pm0 |= 1u << lds_nib(vs, r0); ss0 += (int)(int8_t)vs[F_SOFF + r0]; tq0 = max(tq0, (int)vs[F_QOFF + r0]); }
                    if (in1) { pm1 |= 1u << lds_nib(vs, r1); ss1 += (int)(int8_t)vs[F_SOFF + r1]; tq1 = max(tq1, (int)vs[F_QOFF + r1]); }
                }
            }
            bool c0_ = false, c1_ = false;
            if (a0) {
                const uint8_t ob = tslot[bi];
                resb[col0] = ob >> 4; resq[col0] = (uint8_t)tq0;
                c0_ = !(__popc(pm0) == 1 && ss0 >= accept_score && tq0 >= p.moderate_q);
                if (a1) { resb[col0 + 1] = ob & 0xF; resq[col0 + 1] = (uint8_t)tq1; c1_ = !(__popc(pm1) == 1 && ss1 >= accept_score && tq1 >= p.moderate_q); }
            }
            const unsigned long long m0 = __ballot(c0_), m1 = __ballot(c1_);
            if (c0_) cplx[n_cplx + lanes_below(m0)] = (uint16_t)col0;
            n_cplx += __popcll(m0);
            if (c1_) cplx[n_cplx + lanes_below(m1)] = (uint16_t)(col0 + 1);
            n_cplx += __popcll(m1);
        }
        WAVE_SYNC();
        // ---- pass B: contested columns, one lane each
        int minc = 0;
#pragma unroll 1
        for (int base = 0; base < n_cplx; base += 64) {
            const bool actv = base + lane < n_cplx;
            const int col = actv ? cplx[base + lane] : 0;
            Tally5 t; tally_clear(t);
            for (unsigned long long m = vmask; m; m &= m - 1) {
                const int v = __ffsll((long long)m) - 1;
                const uint8_t *vs = LD + v * F_RS;
                const int vld = left_mode ? 0 : rl32(ld, v), vlq = rl32(lq, v);
                const int rp = col + vld;
                if (actv && rp >= 0 && rp < vlq) tally_add(t, lds_nib(vs, rp), vs[F_QOFF + rp], (int)(int8_t)vs[F_SOFF + rp]);
            }
            if (actv) {
                int ref4 = 0;
                if (ref) {
                    const int ro = (o_nc == 1 && cig_op(o_c0) == 0) ? (col < cig_len(o_c0) ? col : -1) : d_ref_offset(ocig, o_nc, col);
                    if (ro >= 0 && (int64_t)o_pos + ro < ref_n) ref4 = d_ref_nib(ref, (int64_t)o_pos + ro);
                }
                ColOut r = decide_column(t, p, resb[col], ref4);
                resb[col] = (uint8_t)r.base; resq[col] = (uint8_t)r.qual; minc += r.minc;
            }
        }
        WAVE_SYNC();
        minc = wave_sum(minc);
        bool restore = false;
        if (minc != 0) {                                                          // group.cpp:528-573
            if (b.nm_type[out] == 0) { if (lane == 0) raise_error(w.si, GCE_ERR_NM_MISSING, out); restore = true; }
            else if (minc > 5) restore = true;
            else if (lane == 0) { int nn = b.nm[out] + minc; if (b.nm_type[out] == 'C' && nn >= 0 && nn <= 255) w.nm_new[out] = nn; }
        }
        // ---- write the template back: voted bases/quals for columns < len, its rescored quals everywhere else (and on restore)
        uint8_t *oseq = b.seq + o_so, *oqual = b.qual + o_qo;
        for (int i = lane; i < o_lq; i += 64) oqual[i] = (!restore && i < len) ? resq[i] : tslot[F_QOFF + i];
        if (!restore) {
            for (int bi = lane; bi < nbytes; bi += 64) {
                const int c0i = 2 * bi;
                const uint8_t ob = tslot[bi];
                const uint8_t nb = (uint8_t)((resb[c0i] << 4) | ((c0i + 1 < len) ? resb[c0i + 1] : (ob & 0xF)));
                if (nb != ob) oseq[bi] = nb;
            }
        }
        if (lane == 0) rp_out[gi] = out;
        WAVE_SYNC();
    }
}
