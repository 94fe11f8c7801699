// gce_device.hpp — device-side data views and scalar helpers of the MI355X consensus engine.
// gfx950 only (wave64).  Reference citations are relative to /root/reference/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gencore_amd.h"

#define GCE_WAVE 64
#define NONE32 0xFFFFFFFFu
#define EMPTY64 0xFFFFFFFFFFFFFFFFull

// thr_mode of a cluster instance (quirks Q1/Q2)
enum : uint32_t { THR_PROPER = 0, THR_UNPROPER = 1, THR_NEVER = 2 };
// read class
enum : uint8_t { CLS_DROP = 0, CLS_CLUSTERED = 1, CLS_BYPASS = 2 };

struct DevBatch {
    int64_t n;
    const gce_core *core;
    const uint64_t *qname_off; const char *qname;
    const uint64_t *cigar_off; const uint32_t *cigar;
    const uint64_t *seq_off;   uint8_t *seq;
    const uint64_t *qual_off;  uint8_t *qual;
    const int32_t *nm; const uint8_t *nm_type;
    const uint64_t *mi_off; const char *mi;
    const uint64_t *tick;             // optional: global tick of every clustered read (key-range shards), else nullptr
};

struct DevParams {
    int32_t proper_thr, unproper_thr, duplex_mismatch_thr, cluster_size_req, base_score_req;
    int32_t high_q, moderate_q, low_q;
    int32_t s_high, s_moderate, s_low, s_bad;
    int32_t skip_low_complexity_thr, duplex_only, disable_duplex, period;
    int32_t score_bias, score_max;    // score bytes are stored as score + score_bias (>= 0); score_max = largest possible score
    uint32_t q2s_lut;                 // (s_bad, s_low, s_moderate, s_high) + score_bias, one byte each: index = #thresholds passed
    uint32_t thr_low4, thr_mod4, thr_high4;   // the three quality thresholds replicated in 4 bytes
    int32_t q2s_swar_ok;              // thresholds nested and <= 127: qual2score of 4 packed quals in ~13 VALU ops
    double score_percent_req;
    char prefix[32];
    int32_t prefix_len;
    int32_t n_targets;
    const uint32_t *target_len;
    const uint64_t *target_cum;       // exclusive prefix sum of target_len (genome-linear coordinate of each contig)
    int32_t key_bt, key_bl;           // bits of the largest tid / contig length: the packed cluster key of the clustering scan
    int32_t nw_ok, nw_cb, nw_bd;      // normal bucket words (gce_cluster.hpp): usable at all; bits of the read count; bits of right - left + 1
    int32_t vote_ok, vote_accept_by_qual, s_min_lb;   // gce_vote.hpp: score constants in range; "top quality >= moderate" implies "score sum >= baseScoreReq"; smallest score
    int64_t tick_offset;
    int64_t tick_epoch0; int32_t tick_rem0;   // tick_offset / period, tick_offset % period (host side: no 64-bit division in the scan)
    int32_t trailing_flush;
    int32_t n_ref;
    const uint8_t *const *ref_data;   // [n_ref] device pointers or nullptr
    const int64_t *ref_len;           // [n_ref]
    const int64_t *ref_win;           // [2 x n_ref] staged window [lo, hi) of every contig (gce_set_reference_window), or nullptr: whole contigs
};

// stream-level scalars produced by the prescan (device resident)
#define GCE_PRE_SLOTS 64
struct StreamInfo {
    unsigned long long n_clustered;      // number of clustered reads (ticks)
    unsigned int first_unmapped;         // index of the first unmapped read (U) or NONE32
    unsigned long long err_key;          // min over raised errors of (read index << 8 | -gce_status); ~0 = none
    int n_events;                        // E: flush events inside this slice
    int n_events_a;                      // E_A: events whose read index < U
    unsigned int n_slow;                 // group sides deferred to the generic consensus kernel
    unsigned long long prof[32];         // -DVB_PROF builds only: accumulated phase times of k_vote (tools/vote_prof.sh)
    unsigned int n_deep;                 // of those: deep sides prepared for k_vote_deep
    unsigned long long hand_on;          // what k_vote hands on, ONE 64-bit counter bumped once per handed-on group: low half = pair slots on k_score2's list (score_list), high half = sides on
                                         // gen_list (round 5: lists appended to by the group's lane; a flag per slot / per side before, cleared, scanned and compacted every step)
    unsigned int n_slow_pair;            // clusters deferred to the generic pairing kernel
    unsigned int n_slow_pair2;           // of those: left to the generic kernels by k_pairing_deep (pq_list)
    unsigned int pair_next, pair_next2;  // ... and of k_pairing_deep (LDS / device-memory instantiation)
    unsigned int prep_next, deep_next;   // work counters of k_deep_prepare / k_vote_deep: a wave / a block draws its next side when it is done with one
    int lq_min, lq_max;                  // shortest / longest read that can be emitted (clustered or passed through): k_describe
    unsigned long long n_clusters, n_groups, n_pairs, n_out;
    unsigned long long n_pairs_total;    // pairs over all processed clusters
    unsigned long long vote_weight;      // sum of the group weights: k_vote runs vote_weight / VB_W + 1 batches
    unsigned long long n_leaders;        // (cluster, scan block) runs of the clustering scan
    unsigned long long out_units;        // size of the compact output blobs in 16-byte units: bases << 32 | qualities
    unsigned long long n_pf_items;       // clusters the half-wave pairing kernel handed to the full-wave one
    unsigned long long n_pq_items;       // clusters the quarter-wave pairing kernel handed to the half-wave one
    unsigned long long n_p16_items;      // clusters of <= 16 reads: the quarter-wave pairing kernel's list
    long long pre[GCE_STATS_WORDS];
    long long post[GCE_STATS_WORDS];
    long long post_slot[GCE_PRE_SLOTS][8];  // k_out_meta's six addRead counters of the emitted records, spread the same way
    long long pre_slot[GCE_PRE_SLOTS][8];   // k_prescan's six addRead counters, spread over GCE_PRE_SLOTS address sets (block & mask) and added
                                           // up by the host: tens of thousands of blocks adding to SIX words queued behind each other
};

// Fatal conditions of the path: the reference exits on the first one it meets; the engine reports the one on the EARLIEST read
// (deterministic whatever the scheduling: one 64-bit atomicMin on read index << 8 | code).
__device__ __forceinline__ void raise_error(StreamInfo *si, int code, uint32_t read) {
    atomicMin(&si->err_key, ((unsigned long long)read << 8) | (unsigned long long)((-code) & 0xFF));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------------------------------- CIGAR
__device__ __forceinline__ int cig_op(uint32_t w) { return (int)(w & 0xF); }
__device__ __forceinline__ int cig_len(uint32_t w) { return (int)(w >> 4); }
// bam_cigar_type bit0 = consumes query, bit1 = consumes reference; ops 10..15 consume nothing (bamutil.cpp:290-291)
__device__ __forceinline__ int consumes_query(int op) { return (0x193 >> op) & 1; }   // M I S = X  -> bits 0,1,4,7,8
__device__ __forceinline__ int consumes_ref(int op) { return (0x18D >> op) & 1; }     // M D N = X  -> bits 0,2,3,7,8

// Per-read descriptor written once by the thread-per-read prescan: everything the per-pair / per-group kernels need about a
// read in ONE 32-byte record (two 16-byte loads) instead of the dependent chain core -> offsets -> first CIGAR word.
// In memory (ReadDescP): 40-bit blob offsets, first CIGAR word, position with "isize != 0" in the sign bit (clustered reads are mapped:
// pos >= 0), read length, first M block, op count -- 16-bit fields, reads longer than 65535 bases are rejected by the prescan -- and the
// contig: the template of a side needs it for the reference lookup, and fetching it from the core record there was one more dependent
// round trip in a single-wave phase of k_vote.
// Round 1 kept 48 bytes (isize, reference length as well): 320 MB of writes and as many reads per 20 M reads for fields that follow from
// the first CIGAR word (reference length of a one-op read) or of which one bit is used.
struct __attribute__((aligned(16))) ReadDescP {
    uint32_t so_lo, qo_lo;
    uint32_t c0;
    uint32_t pos_fl;          // pos | (isize != 0) << 31
    uint16_t lq, mo, ml, nc;
    int32_t tid;
    uint32_t hi;              // so >> 32 | (qo >> 32) << 8   (blobs live in HBM: offsets < 2^40) | lastm << 16
};
static_assert(sizeof(ReadDescP) == 32, "ReadDescP must stay 32 bytes");
struct ReadDesc {             // the unpacked form the kernels work with
    uint64_t so, qo;          // byte offsets of the packed bases / quals
    uint32_t c0;              // first CIGAR word (0 if none)
    int32_t pos, lq, isize;   // isize: only "is it zero" survives (1 / 0)
    int32_t mo, ml;           // BamUtil::getMOffsetAndLen: first M block (bamutil.cpp:316-336)
    int32_t tid;
    uint16_t nc, lastm;       // n_cigar; length of the LAST CIGAR op if it is an M block (capped at 65535), else 0: what isPartOf looks at
                              // first when reads are aligned at their right end (bamutil.cpp:204-255)
    int32_t rlen;             // bam_cigar2rlen of a ONE-op read, else RLEN_WALK
};
#define RLEN_WALK (-0x40000000)     // more than one CIGAR op: d_cigar_rlen over the CIGAR (desc_rlen)
__device__ __forceinline__ void store_desc(ReadDescP *base, uint64_t i, uint64_t so, uint64_t qo, uint32_t c0, int32_t pos, bool isize_nz, int lq, int mo, int ml, int nc, int32_t tid, uint32_t last_op) {
    union { ReadDescP d; uint4 q[2]; } u;
    u.d.so_lo = (uint32_t)so; u.d.qo_lo = (uint32_t)qo; const uint32_t lastm = cig_op(last_op) == 0 ? min((uint32_t)cig_len(last_op), 65535u) : 0u;
    u.d.hi = ((uint32_t)(so >> 32) & 0xFFu) | ((uint32_t)(qo >> 32) & 0xFFu) << 8 | lastm << 16;
    u.d.c0 = c0; u.d.pos_fl = (uint32_t)pos | (isize_nz ? 0x80000000u : 0u);
    u.d.lq = (uint16_t)lq; u.d.mo = (uint16_t)mo; u.d.ml = (uint16_t)ml; u.d.nc = (uint16_t)nc; u.d.tid = tid;
    uint4 *dst = reinterpret_cast<uint4 *>(base + i); dst[0] = u.q[0]; dst[1] = u.q[1];
}
__device__ __forceinline__ ReadDesc load_desc(const ReadDescP *base, uint32_t i) {
    union { ReadDescP d; uint4 q[2]; } u;
    const uint4 *src = reinterpret_cast<const uint4 *>(base + i);
    u.q[0] = src[0]; u.q[1] = src[1];
    ReadDesc r;
    r.so = (uint64_t)u.d.so_lo | (uint64_t)(u.d.hi & 0xFFu) << 32; r.qo = (uint64_t)u.d.qo_lo | (uint64_t)((u.d.hi >> 8) & 0xFFu) << 32;
    r.c0 = u.d.c0; r.pos = (int32_t)(u.d.pos_fl & 0x7FFFFFFFu); r.isize = (int32_t)(u.d.pos_fl >> 31);
    r.lq = u.d.lq; r.mo = u.d.mo; r.ml = u.d.ml; r.nc = u.d.nc; r.tid = u.d.tid; r.lastm = (uint16_t)(u.d.hi >> 16);
    r.rlen = u.d.nc == 1 ? (int32_t)(cig_len(u.d.c0) * consumes_ref(cig_op(u.d.c0))) : (u.d.nc == 0 ? 0 : RLEN_WALK);
    return r;
}


// BamUtil::isPartOf (bamutil.cpp:204-255): op-by-op containment, from the end when right aligned.
__device__ inline bool d_is_part_of(const uint32_t *part, int np, const uint32_t *whole, int nw, bool left) {
    if (nw < np) return false;
    for (int i = 0; i < np; i++) {
        uint32_t a = left ? part[i] : part[np - 1 - i];
        uint32_t b = left ? whole[i] : whole[nw - 1 - i];
        if (cig_op(a) != cig_op(b)) return false;
        int la = cig_len(a), lb = cig_len(b);
        if (la > lb) return false;
        if (la < lb && i != np - 1) {
            if (i != np - 2) return false;
            uint32_t nx = left ? part[i + 1] : part[np - 2 - i];
            if (cig_op(nx) != 5 /*H*/) return false;
        }
    }
    return true;
}

// BamUtil::getRefOffset (bamutil.cpp:293-314)
__device__ inline int d_ref_offset(const uint32_t *cig, int n, int qpos) {
    int ref = 0, query = 0;
    for (int i = 0; i < n; i++) {
        int op = cig_op(cig[i]), len = cig_len(cig[i]);
        int cq = consumes_query(op), cr = consumes_ref(op);
        query += len * cq;
        ref += len * cr;
        if (query > qpos) {
            if (op == 1 || op == 4) return -1;           // inside I or S
            return ref - cr * (query - qpos);
        }
    }
    return -1;
}

// BamUtil::getMOffsetAndLen (bamutil.cpp:316-336): first M block only
__device__ inline void d_first_m(const uint32_t *cig, int n, int &off, int &len) {
    int query = 0;
    for (int i = 0; i < n; i++) {
        int op = cig_op(cig[i]), l = cig_len(cig[i]);
        if (op == 0) { off = query; len = l; return; }
        query += l * consumes_query(op);
    }
    off = 0; len = 0;
}

// bam_cigar2rlen (used by BamUtil::getRightRefPos, bamutil.cpp:379-383)
__device__ inline int d_cigar_rlen(const uint32_t *cig, int n) {
    int l = 0;
    for (int i = 0; i < n; i++) l += cig_len(cig[i]) * consumes_ref(cig_op(cig[i]));
    return l;
}

// ---------------------------------------------------------------------------------------------------- bases
__device__ __forceinline__ int d_nib(const uint8_t *s, int i) {
    uint8_t b = s[i >> 1];
    return (i & 1) ? (b & 0xF) : (b >> 4);
}
// BamUtil::fourbits2base collapsed to an equivalence class: 1,2,4,8 keep their identity, everything else is 'N'
// (bamutil.cpp:148-165).  Two nibbles decode to the same char iff their classes are equal.
__device__ __forceinline__ int d_base_class(int nib4) { return (nib4 == 1 || nib4 == 2 || nib4 == 4 || nib4 == 8) ? nib4 : 15; }

// FastaReader::getBase (fastareader.cpp:122-128) folded with group.cpp:438-439 and BamUtil::base2fourbits:
// returns the BAM nibble of the reference base (1,2,4,8) or 0 when the base is not A/T/C/G.
__device__ __forceinline__ int d_ref_nib(const uint8_t *ref, int64_t pos) {
    int b = ref[pos >> 1];
    int code = (pos & 1) ? (b >> 4) : (b & 0xF);     // FASTA code: A=1,T=2,C=3,G=4
    // -> BAM nibble: A=1, T=8, C=2, G=4
    return (int)((0x42810u >> (4 * (code < 5 ? code : 5))) & 0xFu);   // nibble table {0,1,8,2,4}, 0 beyond
}

// Pair::qual2score (pair.cpp:77-86)
// (round 6) With nested thresholds (q2s_swar_ok: low <= moderate <= high <= 127, every sane configuration) the NUMBER of thresholds passed indexes the four scores packed in
// q2s_lut -- pure ALU.  The nested selects below compile to a select of ADDRESSES into the kernel-argument segment and one vector load from it: a dependent trip to
// memory per score wherever a kernel scores base by base (k_vote's items until round 6, k_score2's overlap units, the bytewise paths of the deep kernels).
__device__ __forceinline__ int d_qual2score(const DevParams &p, int q) {
    if (p.q2s_swar_ok) {
        const int nthr = (q >= p.low_q) + (q >= p.moderate_q) + (q >= p.high_q);
        return (int)((p.q2s_lut >> (8 * nthr)) & 0xFFu) - p.score_bias;
    }
    return q >= p.high_q ? p.s_high : q >= p.moderate_q ? p.s_moderate : q >= p.low_q ? p.s_low : p.s_bad;
}

// Pair::qual2score on four packed quals (each < 128): count the thresholds passed per byte, then one byte-permute through
// the 4-entry score table.  Result bytes are score + score_bias.
__device__ __forceinline__ uint32_t d_q2s4_biased(const DevParams &p, uint32_t q4) {
    if (p.q2s_swar_ok) {
        const uint32_t t = q4 | 0x80808080u;
        const uint32_t c = (((t - p.thr_low4) & 0x80808080u) >> 7) + (((t - p.thr_mod4) & 0x80808080u) >> 7) + (((t - p.thr_high4) & 0x80808080u) >> 7);
        return __builtin_amdgcn_perm(0u, p.q2s_lut, c);
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) r |= (uint32_t)((d_qual2score(p, (q4 >> (8 * k)) & 0xFF) + p.score_bias) & 0xFF) << (8 * k);
    return r;
}
#define GCE_PATCH_CONST 0xFFFFFFFFu
// Score of base `pos` of a read (pair.cpp:88-172): constant for a read scored without a usable mate, the stored overlap patch
// inside [start, start+len), qual2score(qual) everywhere else.  patch = start | len << 16.
__device__ __forceinline__ int d_score_at(const DevParams &p, const int8_t *score_row, uint32_t patch, int pos, int q) {
    if (patch == GCE_PATCH_CONST) return p.s_moderate;
    if ((unsigned)(pos - (int)(patch & 0xFFFF)) < (patch >> 16)) return (int)(uint8_t)score_row[pos] - p.score_bias;
    return d_qual2score(p, q);
}

// ---------------------------------------------------------------------------------------------------- UMI
__device__ __forceinline__ bool d_is_umi_char(char c) { return c == 'A' || c == 'T' || c == 'C' || c == 'G' || c == '_'; }

// BamUtil::getUMI(string, prefix) (bamutil.cpp:40-112).  Returns false where the reference throws.
// `n` = strlen(s) when the caller already knows it (l_qname - 1), or -1.
// ---- packed-byte helpers for the name parsers: masks carry 0x80 in every selected byte of a 64-bit word
typedef uint64_t u64_unaligned_t __attribute__((aligned(1)));
__device__ __forceinline__ uint64_t d_zero_bytes(uint64_t x) {          // exact: 0x80 where the byte of x is zero
    const uint64_t m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((x & m) + m) | x | m);
}
__device__ __forceinline__ uint64_t d_eq_bytes(uint64_t w, char c) { return d_zero_bytes(w ^ (0x0101010101010101ull * (uint8_t)c)); }
__device__ __forceinline__ uint64_t d_umi_bytes(uint64_t w) {            // [ATCG_]
    return d_eq_bytes(w, 'A') | d_eq_bytes(w, 'T') | d_eq_bytes(w, 'C') | d_eq_bytes(w, 'G') | d_eq_bytes(w, '_');
}
// bytes lo <= j < hi of word k of the window (j = 8k + byte), as a full-byte mask
__device__ __forceinline__ uint64_t d_range_bytes(int lo, int hi, int k) {
    const int a = min(max(lo - 8 * k, 0), 8), z = min(max(hi - 8 * k, 0), 8);
    const uint64_t upto_z = z >= 8 ? ~0ull : ((1ull << (8 * z)) - 1ull), upto_a = a >= 8 ? ~0ull : ((1ull << (8 * a)) - 1ull);
    return upto_z & ~upto_a;
}

__device__ inline bool d_umi_slice_bytes(const char *s, const DevParams &p, int &start, int &len, int n);

// BamUtil::getUMI(string, prefix) on the LAST 32 BYTES of a name held in four registers (bamutil.cpp:40-112): the prefix
// character / ':' that anchors the UMI is practically always inside that window, and the scans become a handful of packed
// compares instead of two divergent byte loops per read.  Anything the window cannot answer takes the byte walk.
// Reads up to 7 bytes past s[n-1]: device blobs are readable 16 bytes past their end (include/gencore_amd.h).
__device__ inline bool d_umi_slice(const char *s, const DevParams &p, int &start, int &len, int n = -1) {
    if (n < 0 || p.prefix_len > 4) return d_umi_slice_bytes(s, p, start, len, n);
    const int base = n > 32 ? n - 32 : 0, nwin = n - base;
    uint64_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = 8 * k < nwin ? *(const u64_unaligned_t *)(s + base + 8 * k) : 0ull;
    start = 0; len = 0;
    // last prefix character / ':' : words are examined from the end and only while nothing was found (the anchor of a real
    // name sits in the last word or two, so the earlier words usually cost nothing)
    int wpos = -1;                                                         // window index of the last anchor byte
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        if (wpos < 0 && 8 * k < nwin) {
            uint64_t h = 0;
            if (p.prefix_len > 0) { for (int c = 0; c < p.prefix_len; c++) h |= d_eq_bytes(w[k], p.prefix[c]); }
            else h = d_eq_bytes(w[k], ':');
            h &= d_range_bytes(0, nwin, k);
            if (h) wpos = 8 * k + ((63 - __clzll((long long)h)) >> 3);
        }
    }
    if (wpos < 0) {
        if (base == 0) return true;                                        // no anchor anywhere: no UMI
        return d_umi_slice_bytes(s, p, start, len, n);                     // anchor (if any) in front of the window
    }
    if (p.prefix_len > 0) {
        const int wst = wpos + 2;                                          // find_last_of + 2 (bamutil.cpp:47-50)
        if (base + wst > n) return false;                                  // substr(start) with start > size() throws
        int l = nwin - wst;                                                // run of [ATCG_] from wst, ended by the first other byte
        bool open = true;                                                  // (forward from the word that holds wst; stop at the first hit)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (open && 8 * k + 8 > wst && 8 * k < nwin) {
                const uint64_t stop = ~d_umi_bytes(w[k]) & 0x8080808080808080ull & d_range_bytes(wst, nwin, k);
                if (stop) { l = 8 * k + ((__ffsll((long long)stop) - 1) >> 3) - wst; open = false; }
            }
        }
        start = base + wst; len = l;
        return true;
    }
    // no prefix: text after the last ':' if it is all [ATCG_] with at most one '_' (bamutil.cpp:65-111)
    const int sep = base + wpos;
    if (sep >= n - 1) return true;
    bool ok = true; int us = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t tail = d_range_bytes(wpos + 1, nwin, k);
        if (~d_umi_bytes(w[k]) & 0x8080808080808080ull & tail) ok = false;
        us += __popcll(d_eq_bytes(w[k], '_') & tail);
    }
    int st = sep + 1;
    { const int j = wpos + 1; const char c = (char)(w[j >> 3] >> (8 * (j & 7))); if (st < n - 1 && c == '_') { st++; us--; } }
    if (!ok || us > 1) return true;
    start = st; len = n - st;
    return true;
}

__device__ inline bool d_umi_slice_bytes(const char *s, const DevParams &p, int &start, int &len, int n) {
    if (n < 0) { n = 0; while (s[n]) n++; }
    start = 0; len = 0;
    if (p.prefix_len > 0) {
        int pos = -1;
        for (int i = n - 1; i >= 0; i--) {
            char c = s[i];
            bool hit = false;
            for (int k = 0; k < p.prefix_len; k++) hit |= (p.prefix[k] == c);
            if (hit) { pos = i; break; }
        }
        if (pos < 0) return true;
        int st = pos + 2;
        if (st > n) return false;                      // substr(start) with start > size() throws
        // every character after the last prefix character up to the end was examined by the backward scan; of those only a
        // run of [ATCG_] right after st counts
        int l = 0;
        while (st + l < n && d_is_umi_char(s[st + l])) l++;
        start = st; len = l;
        return true;
    }
    int sep = -1, us = 0;
    bool ok = true;
    for (int i = n - 1; i >= 0; i--) {                   // one backward pass: find the last ':' and validate the tail on the way
        char c = s[i];
        if (c == ':') { sep = i; break; }
        if (!d_is_umi_char(c)) ok = false;
        if (c == '_') us++;
    }
    if (sep < 0 || sep >= n - 1) return true;
    int st = sep + 1;
    if (st < n - 1 && s[st] == '_') { st++; us--; }      // one leading underscore is skipped, and not counted
    if (!ok || us > 1) return true;
    start = st; len = n - st;
    return true;
}

// Cluster::umiDiff (cluster.cpp:41-53)
__device__ inline int d_umi_diff(const char *a, int la, const char *b, int lb) {
    int m = la < lb ? la : lb;
    int d = la > lb ? la - lb : lb - la;
    for (int i = 0; i < m; i++) d += (a[i] != b[i]);
    return d;
}
__device__ inline bool d_bytes_equal(const char *a, int la, const char *b, int lb) {
    if (la != lb) return false;
    for (int i = 0; i < la; i++) if (a[i] != b[i]) return false;
    return true;
}
// std::string operator< on slices
__device__ inline int d_slice_cmp(const char *a, int la, const char *b, int lb) {
    int m = la < lb ? la : lb;
    for (int i = 0; i < m; i++) {
        unsigned char x = (unsigned char)a[i], y = (unsigned char)b[i];
        if (x != y) return x < y ? -1 : 1;
    }
    return la - lb;
}
__device__ inline int d_strcmp(const char *a, const char *b) {
    for (int i = 0;; i++) {
        unsigned char x = (unsigned char)a[i], y = (unsigned char)b[i];
        if (x != y) return x < y ? -1 : 1;
        if (x == 0) return 0;
    }
}

// util.h:59-88 split(str, "_") reduced to what Cluster::isDuplex (cluster.cpp:246-258) needs: the token count and
// the first two tokens.  Leading underscores are skipped, every later '_' closes a token (possibly an empty one),
// and a trailing '_' yields one more empty token.
__device__ inline int d_split2(const char *s, int n, int &s0, int &l0, int &s1, int &l1) {
    int i = 0;
    while (i < n && s[i] == '_') i++;
    if (i >= n) return 0;
    int cnt = 0, ts = i;
    s0 = l0 = s1 = l1 = 0;
    for (;; ) {
        int j = ts;
        while (j < n && s[j] != '_') j++;
        if (cnt == 0) { s0 = ts; l0 = j - ts; }
        else if (cnt == 1) { s1 = ts; l1 = j - ts; }
        cnt++;
        if (j >= n) break;                               // no further separator
        ts = j + 1;                                      // may equal n: one more, empty, token
    }
    return cnt;
}
__device__ inline bool d_is_duplex(const char *a, int la, const char *b, int lb) {
    int a0, al0, a1, al1, b0, bl0, b1, bl1;
    if (d_split2(a, la, a0, al0, a1, al1) != 2) return false;
    if (d_split2(b, lb, b0, bl0, b1, bl1) != 2) return false;
    return d_bytes_equal(a + a0, al0, b + b1, bl1) && d_bytes_equal(a + a1, al1, b + b0, bl0);
}

// ---------------------------------------------------------------------------------------------------- cluster key
struct ClusterKey { int32_t tid, left; int64_t right; };

// Gencore::addToProperCluster key derivation (gencore.cpp:295-312).  Returns the read class.
__device__ __forceinline__ uint8_t d_classify(const gce_core &c) {
    if (c.tid < 0 || c.pos < 0) return CLS_DROP;                      // unmapped: counted, then dropped (gencore.cpp:255-266)
    if (c.flag & (0x100 | 0x800)) return CLS_DROP;                    // secondary / supplementary (gencore.cpp:269-271)
    long long d = (long long)c.mpos - (long long)c.pos;
    if (d < 0) d = -d;
    if (c.mtid == c.tid && d < 100000) return CLS_CLUSTERED;
    if (c.mtid < 0) return CLS_BYPASS;                                 // mate unmapped: written as is (gencore.cpp:307-309)
    return CLS_CLUSTERED;
}
__device__ __forceinline__ ClusterKey d_key(const gce_core &c, const DevParams &p) {
    ClusterKey k;
    k.tid = c.tid; k.left = c.pos;
    long long d = (long long)c.mpos - (long long)c.pos;
    if (d < 0) d = -d;
    if (c.mtid == c.tid && d < 100000) {
        if (c.isize < 0) k.left = c.mpos;
        long long a = c.isize; if (a < 0) a = -a;
        k.right = (long long)k.left + a - 1;
    } else {
        long long tl = (c.tid < p.n_targets && p.target_len) ? (long long)p.target_len[c.tid] : 0;
        k.right = -1LL * tl * (long long)(c.mtid + 1) + (long long)c.mpos;
    }
    return k;
}
// x mod T for a table size that is not a power of two: double-reciprocal quotient estimate (exact after one correction
// step while x < 2^52, which covers every genome-linear bucket index); anything larger takes the 64-bit remainder.
__device__ __forceinline__ void d_divmod(uint64_t x, uint64_t T, double tinv, uint64_t &q, uint64_t &r) {
    if (x >> 52) { q = x / T; r = x - q * T; return; }
    uint64_t qq = (uint64_t)((double)x * tinv);
    long long rr = (long long)(x - qq * T);
    while (rr < 0) { rr += (long long)T; qq--; }
    while (rr >= (long long)T) { rr -= (long long)T; qq++; }
    q = qq; r = (uint64_t)rr;
}
__device__ __forceinline__ uint64_t d_bucket(uint64_t x, uint64_t T, double tinv) {
    if (x >> 52) return x % T;
    const uint64_t q = (uint64_t)((double)x * tinv);
    long long r = (long long)(x - q * T);
    while (r < 0) r += (long long)T;
    while (r >= (long long)T) r -= (long long)T;
    return (uint64_t)r;
}

// padded in-memory l_qname (htslib l_extranul)
__device__ __forceinline__ int d_lqname_pad(const gce_core &c) { return ((int)c.l_qname + 3) & ~3; }

// wave-level helpers
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
// inclusive scan over the 64 lanes of a fully active wave in six DPP adds (row shifts 1, 2, 4, 8 inside the rows of 16, then lane 15 /
// lane 31 broadcast into the following rows): __shfl_up is a ds_bpermute round trip per step, and the scans sit in the single-wave phases
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// The same reductions on the DPP path, for fully active waves: rotations inside the rows of 16 lanes (one VALU op a step, no LDS
// crossbar round trip), the four row results read back as scalars.  The result is wave-uniform (an SGPR).
template <int CTRL> __device__ __forceinline__ int dpp_move(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_max16(int v) {          // max over the lane's row of 16, in every lane of the row
    v = max(v, dpp_move<0x128>(v));                          // row_ror:8
    v = max(v, dpp_move<0x124>(v));                          // row_ror:4
    v = max(v, dpp_move<0x122>(v));                          // row_ror:2
    v = max(v, dpp_move<0x121>(v));                          // row_ror:1
    return v;
}
__device__ __forceinline__ int wave_max_u(int v) {
    const int r = row_max16(v);
    return max(max(__builtin_amdgcn_readlane(r, 0), __builtin_amdgcn_readlane(r, 16)), max(__builtin_amdgcn_readlane(r, 32), __builtin_amdgcn_readlane(r, 48)));
}
__device__ __forceinline__ int lanes_below(unsigned long long mask) {   // popcount of mask bits below this lane
    return __popcll(mask & ((1ull << lane_id()) - 1ull));
}
