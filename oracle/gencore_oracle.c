/*
 * gencore_oracle.c — CPU oracle: a plain-C restatement of the Cluster -> Group -> consensus path of
 * OpenGene/gencore v0.17.2 (reference tree /root/reference, read-only).
 *
 * TEST INFRASTRUCTURE ONLY (see gencore_oracle.h).  PARITY PINNING: get_umi / umi_diff / is_duplex are pinned by
 * the reference's own known-answer vectors, and is_duplex additionally against the reference's own tokenizer
 * (src/util.h is the one file of the path that compiles without htslib: oracle/ref_probe -> oracle/_ref/libref_util.so);
 * everything else is "parity unpinned" (the reference cannot be built here: it needs htslib, which is absent, and
 * stand-in headers are not allowed).
 *
 * The stream driver below SIMULATES the reference literally — a pending-cluster set, a tick counter, a flush
 * walk every `flush_period` clustered reads — rather than using the closed form the HIP engine uses for the
 * same rule, so that the two derivations check each other.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/src).
 */
#include "gencore_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NONE GCE_NONE

/* ------------------------------------------------------------------------------------------------ helpers */

typedef struct OPair {
    uint32_t     left, right;       /* Pair::mLeft / mRight as read indices             pair.h:50-51   */
    const char  *umi;               /* Pair::mUMI (slice of a qname / MI string)         pair.h:61      */
    int          umi_len;
    signed char *lscore, *rscore;   /* Pair::mLeftScore / mRightScore                    pair.h:65-66   */
    int          merge_reads;       /* mMergeReads                                       pair.h:52      */
    int          reverse_merge_reads;
    int          is_duplex;
} OPair;

typedef struct OCluster {
    int32_t   tid, left;
    int64_t   right;
    uint32_t *reads;                /* arrival order */
    int       n, cap;
} OCluster;

typedef struct Ctx {
    const gce_params    *prm;
    const orc_reference *ref;
    gce_batch           *b;
    orc_result          *res;
    OCluster           **pending;   /* the reference's mProperClusters, gencore.h:76 */
    int64_t              n_pending, cap_pending;
    uint32_t            *htab;      /* open-addressing index into pending[] (+1), rebuilt after each flush */
    uint64_t             hmask;
    int64_t              tick;      /* static int tick, gencore.cpp:319 */
    int32_t              n_ev, next_ev; const int32_t *ev_tid, *ev_pos;   /* key-range shards: flush events of the whole stream */
    int                  failed;
} Ctx;

static void fail(Ctx *c, int status, const char *msg) {
    if (c->failed) return;
    c->failed = 1;
    c->res->status = status;
    snprintf(c->res->message, sizeof c->res->message, "%s", msg);
}

static inline const gce_core *core_of(Ctx *c, uint32_t r) { return &c->b->core[r]; }
/* current qname of a record: copyQName (bamutil.cpp:338-366) is tracked as an index indirection */
static inline const char *qname_of(Ctx *c, uint32_t r) { return c->b->qname + c->b->qname_off[c->res->qname_src[r]]; }
/* bam1_core_t.l_qname as htslib keeps it in memory: strlen+1 padded with l_extranul NULs to a multiple of 4 */
static inline int lqname_of(Ctx *c, uint32_t r) { return (core_of(c, c->res->qname_src[r])->l_qname + 3) & ~3; }
static inline const uint32_t *cigar_of(Ctx *c, uint32_t r) { return c->b->cigar + c->b->cigar_off[r]; }
static inline uint8_t *seq_of(Ctx *c, uint32_t r) { return c->b->seq + c->b->seq_off[r]; }
static inline uint8_t *qual_of(Ctx *c, uint32_t r) { return c->b->qual + c->b->qual_off[r]; }
static inline int nib(const uint8_t *s, int i) { return (i & 1) ? (s[i >> 1] & 0xF) : ((s[i >> 1] >> 4) & 0xF); }

/* ------------------------------------------------------------------------------------------ BamUtil pieces */

/* BamUtil::getUMI(string qname, const string& prefix)                                   bamutil.cpp:40-112.
 * Returns the UMI as a slice [*start, *start+*len) of `name`; -1 where the reference would throw
 * (substr with start > length, bamutil.cpp:62). */
static int umi_slice(const char *name, const char *prefix, int *start_out, int *len_out) {
    int len = (int)strlen(name);
    int plen = (int)strlen(prefix);
    *start_out = 0; *len_out = 0;
    if (plen > 0) {                                         /* prefix mode, :45-63 */
        int pos = -1;
        for (int i = len - 1; i >= 0 && pos < 0; i--)       /* find_last_of(prefix): last char that is ANY char of prefix */
            if (strchr(prefix, name[i]) != NULL) pos = i;
        if (pos < 0) return 0;                              /* npos -> "" */
        int start = pos + 2, n = 0;
        for (int s = start; s < len; s++) {
            char ch = name[s];
            if (ch != 'A' && ch != 'T' && ch != 'C' && ch != 'G' && ch != '_') break;
            n++;
        }
        if (start > len) return -1;                         /* std::out_of_range in the reference */
        *start_out = start; *len_out = n;
        return 0;
    }
    int sep = -1;                                           /* no-prefix mode, :65-111 */
    for (int i = len - 1; i >= 0; i--) if (name[i] == ':') { sep = i; break; }
    if (sep < 0 || sep >= len - 1) return 0;                /* :76-79 (prefixLen == 0) */
    int start = sep + 1;
    if (start < len - 1 && name[start] == '_') start++;     /* :94-96 */
    int underscores = 0;
    for (int i = start; i < len; i++) {                     /* :98-110 */
        char ch = name[i];
        if (ch != 'A' && ch != 'T' && ch != 'C' && ch != 'G' && ch != '_') return 0;
        if (ch == '_' && ++underscores > 1) return 0;
    }
    *start_out = start; *len_out = len - start;
    return 0;
}

int orc_get_umi(const char *name, const char *prefix, char *out, int cap) {
    int s, n;
    if (umi_slice(name, prefix, &s, &n) < 0) return -1;
    if (n >= cap) n = cap - 1;
    memcpy(out, name + s, (size_t)n);
    out[n] = 0;
    return n;
}

/* Cluster::umiDiff                                                                        cluster.cpp:41-53 */
int orc_umi_diff(const char *a, int la, const char *b, int lb) {
    int diff = la > lb ? la - lb : lb - la;
    int m = la < lb ? la : lb;
    for (int i = 0; i < m; i++) if (a[i] != b[i]) diff++;
    return diff;
}

/* split(str, out, "_")                                                                    util.h:59-88.
 * Token boundaries only; returns the token count (max 3 recorded: isDuplex only asks "== 2"). */
static int split_us(const char *s, int n, int tok_start[3], int tok_len[3]) {
    int cnt = 0;
    if (n == 0) return 0;
    int pb = 0;
    while (pb < n && s[pb] == '_') pb++;                    /* find_first_not_of */
    if (pb >= n) return 0;
    for (;;) {
        int cp = -1;
        for (int i = pb; i < n; i++) if (s[i] == '_') { cp = i; break; }
        int ts = pb, tl;
        if (cp >= 0) { tl = cp - pb; pb = cp + 1; }
        else         { tl = n - pb;  pb = -1; }
        if (cnt < 3) { tok_start[cnt] = ts; tok_len[cnt] = tl; }
        cnt++;
        if (pb < 0) break;                                  /* pos_begin == npos */
        /* pb may equal n: the reference then pushes one more, empty, token (find returns npos, substr(n) == "") */
    }
    return cnt;
}

/* Cluster::isDuplex                                                                      cluster.cpp:246-258 */
int orc_is_duplex(const char *a, int la, const char *b, int lb) {
    int as[3], al[3], bs[3], bl[3];
    if (split_us(a, la, as, al) != 2 || split_us(b, lb, bs, bl) != 2) return 0;
    return al[0] == bl[1] && al[1] == bl[0] && memcmp(a + as[0], b + bs[1], (size_t)al[0]) == 0 &&
           memcmp(a + as[1], b + bs[0], (size_t)al[1]) == 0;
}

#define CIG_OP(v)  ((int)((v) & 0xF))
#define CIG_LEN(v) ((int)((v) >> 4))
enum { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8 };
/* bamutil.cpp:290-291 (16-entry tables, entries 10..15 zero) */
static const int QUERY_CONSUM[16] = {1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const int REFERENCE_CONSUM[16] = {1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0};

/* BamUtil::isPartOf                                                                      bamutil.cpp:204-255 */
int orc_is_part_of(const uint32_t *part, int n_part, const uint32_t *whole, int n_whole, int is_left) {
    if (n_whole < n_part) return 0;
    for (int i = 0; i < n_part; i++) {
        uint32_t vp = is_left ? part[i] : part[n_part - i - 1];
        uint32_t vw = is_left ? whole[i] : whole[n_whole - i - 1];
        if (CIG_OP(vp) != CIG_OP(vw)) return 0;
        if (CIG_LEN(vp) > CIG_LEN(vw)) return 0;
        if (CIG_LEN(vp) < CIG_LEN(vw)) {
            if (i != n_part - 1) {                          /* shorter only in the last op ... */
                if (i != n_part - 2) return 0;              /* ... or the one before a trailing hard clip */
                int next = i + 1;
                uint32_t vn = is_left ? part[next] : part[n_part - next - 1];
                if (CIG_OP(vn) != C_H) return 0;
            }
        }
    }
    return 1;
}

/* BamUtil::getRefOffset                                                                  bamutil.cpp:293-314 */
int orc_ref_offset(const uint32_t *cigar, int n_cigar, int bampos) {
    int ref = 0, query = 0;
    for (int i = 0; i < n_cigar; i++) {
        int op = CIG_OP(cigar[i]), len = CIG_LEN(cigar[i]);
        query += len * QUERY_CONSUM[op];
        ref += len * REFERENCE_CONSUM[op];
        if (query > bampos) {
            if (op == C_I || op == C_S) return -1;
            return ref - REFERENCE_CONSUM[op] * (query - bampos);
        }
    }
    return -1;                                              /* "wrong cigar" */
}

/* BamUtil::getMOffsetAndLen                                                              bamutil.cpp:316-336 */
void orc_m_offset_len(const uint32_t *cigar, int n_cigar, int *m_off, int *m_len) {
    int query = 0;
    for (int i = 0; i < n_cigar; i++) {
        int op = CIG_OP(cigar[i]), len = CIG_LEN(cigar[i]);
        if (op == C_M) { *m_off = query; *m_len = len; return; }
        query += len * QUERY_CONSUM[op];
    }
    *m_off = 0; *m_len = 0;
}

/* htslib bam_cigar2rlen (used by BamUtil::getRightRefPos, bamutil.cpp:379-383): sum of reference-consuming ops */
int orc_cigar_rlen(const uint32_t *cigar, int n_cigar) {
    int l = 0;
    for (int i = 0; i < n_cigar; i++) {
        int op = CIG_OP(cigar[i]);
        if (op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X) l += CIG_LEN(cigar[i]);
    }
    return l;
}
static int right_ref_pos(Ctx *c, uint32_t r) {
    const gce_core *k = core_of(c, r);
    if (k->pos < 0) return -1;
    return k->pos + orc_cigar_rlen(cigar_of(c, r), k->n_cigar);
}

/* BamUtil::fourbits2base / base2fourbits                                                 bamutil.cpp:148-183 */
static char fourbits2base(int v) {
    switch (v) { case 1: return 'A'; case 2: return 'C'; case 4: return 'G'; case 8: return 'T'; default: return 'N'; }
}
static int base2fourbits(char b) {
    switch (b) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}

/* FastaReader::base2bits + to4bits                                              fastareader.cpp:106-113,139-152 */
void orc_pack_reference(const char *bases, int64_t n, uint8_t *out) {
    memset(out, 0, (size_t)((n + 1) / 2));
    for (int64_t i = 0; i < n; i++) {
        char ch = bases[i];
        uint8_t bits = ch == 'A' ? 1 : ch == 'T' ? 2 : ch == 'C' ? 3 : ch == 'G' ? 4 : 0;
        if ((i & 1) == 0) out[i / 2] |= bits; else out[i / 2] |= (uint8_t)(bits << 4);
    }
}
/* FastaReader::getBase + bits2base                                                fastareader.cpp:115-128 */
char orc_reference_base(const uint8_t *data, int64_t pos) {
    static const char bases[5] = {'N', 'A', 'T', 'C', 'G'};
    int b = data[pos / 2];
    int bits = (pos % 2 == 0) ? (b & 0x0F) : ((b & 0xF0) >> 4);
    return bits >= 5 ? 'N' : bases[bits];
}

/* Reference::getData                                                                     reference.cpp:33-70.
 * The last-contig cache is a pure memo; the observable rule is: NULL if the contig is missing or
 * pos+len >= contig length, else the whole-contig array. */
static const uint8_t *reference_data(Ctx *c, int tid, int64_t pos, int64_t len, int64_t *n_bases) {
    if (!c->ref || tid < 0 || tid >= c->ref->n_contigs) return NULL;
    if (!c->ref->data[tid]) return NULL;
    if (pos + len >= c->ref->n_bases[tid]) return NULL;
    *n_bases = c->ref->n_bases[tid];
    return c->ref->data[tid];
}

/* ------------------------------------------------------------------------------------------------- Stats */

/* Stats::addRead (scalar counters)                                                         stats.cpp:101-121 */
static void stats_add_read(gce_stats *s, const gce_core *k, int has_nm, int nm) {
    int mapped = k->tid >= 0;
    int mismatch = (mapped && has_nm) ? nm : 0;             /* BamUtil::getED, bamutil.cpp:124-131 */
    s->bases += k->l_qseq;
    s->reads++;
    s->base_mismatches += mismatch;
    if (!mapped) { s->bases_unmapped += k->l_qseq; s->reads_unmapped++; }
    if (mismatch > 0) s->reads_with_mismatches++;
}
/* Stats::addMolecule                                                                       stats.cpp:123-133 */
static void stats_add_molecule(gce_stats *s, unsigned supporting, int pe) {
    s->molecules++;
    if (supporting < GCE_MAX_SUPPORTING_READS) s->supporting_hist[supporting]++;
    else s->uncounted_supporting_reads++;
    if (pe) s->molecules_pe++; else s->molecules_se++;
}
/* Stats::addCluster                                                                        stats.cpp:135-139 */
static void stats_add_cluster(gce_stats *s, int multi) { s->clusters++; if (multi) s->multi_molecule_clusters++; }

/* -------------------------------------------------------------------------------------------------- Pair */

/* BamUtil::getUMI(const bam1_t*, prefix)                                                  bamutil.cpp:23-38:
 * MI:Z aux if present, else the (current) qname. */
static int read_umi(Ctx *c, uint32_t r, const char **p, int *n) {
    const char *src;
    if (c->b->mi && c->b->mi_off && c->b->mi_off[r] != UINT64_MAX) src = c->b->mi + c->b->mi_off[r];
    else src = qname_of(c, r);
    int s, l;
    if (umi_slice(src, c->prm->umi_prefix, &s, &l) < 0) { fail(c, GCE_ERR_UMI_PARSE, "UMI substr out of range"); *p = src; *n = 0; return -1; }
    *p = src + s; *n = l;
    return 0;
}

/* Pair::setLeft                                                                             pair.cpp:188-193 */
static void pair_set_left(Ctx *c, OPair *p, uint32_t r) {
    p->left = r;
    read_umi(c, r, &p->umi, &p->umi_len);
}
/* Pair::setRight                                                                            pair.cpp:195-216 */
static void pair_set_right(Ctx *c, OPair *p, uint32_t r) {
    p->right = r;
    const char *u; int n;
    read_umi(c, r, &u, &n);
    if (p->umi_len != 0 && (n != p->umi_len || memcmp(u, p->umi, (size_t)n) != 0))
        fail(c, GCE_ERR_UMI_MISMATCH, "The UMI of a read pair should be identical");
    else { p->umi = u; p->umi_len = n; }
}

/* Pair::qual2score                                                                            pair.cpp:77-86 */
static signed char qual2score(const gce_params *o, uint8_t q) {
    if (o->high_quality <= q) return (signed char)o->score_high;
    if (o->moderate_quality <= q) return (signed char)o->score_moderate;
    if (o->low_quality <= q) return (signed char)o->score_low;
    return (signed char)o->score_bad;
}
/* Pair::assignNonOverlappedScores                                                             pair.cpp:70-75 */
static void assign_scores(const gce_params *o, const uint8_t *qual, int start, int end, signed char *scores) {
    for (int i = start; i < end; i++) scores[i] = qual2score(o, qual[i]);
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Pair::computeScore                                                                         pair.cpp:88-172 */
static void pair_compute_score(Ctx *c, OPair *p) {
    const gce_params *o = c->prm;
    if (p->left != NONE && !p->lscore) {
        int n = core_of(c, p->left)->l_qseq;
        p->lscore = (signed char *)malloc((size_t)(n > 0 ? n : 1));
        memset(p->lscore, o->score_moderate, (size_t)n);
    }
    if (p->right != NONE && !p->rscore) {
        int n = core_of(c, p->right)->l_qseq;
        p->rscore = (signed char *)malloc((size_t)(n > 0 ? n : 1));
        memset(p->rscore, o->score_moderate, (size_t)n);
    }
    if (!(p->lscore && p->rscore)) return;
    const gce_core *lk = core_of(c, p->left), *rk = core_of(c, p->right);
    int lmo, lml, rmo, rml;
    orc_m_offset_len(cigar_of(c, p->left), lk->n_cigar, &lmo, &lml);
    orc_m_offset_len(cigar_of(c, p->right), rk->n_cigar, &rmo, &rml);
    if (!(lml > 0 && rml > 0)) return;
    int pos_dis = rk->pos - lk->pos;
    int lstart, rstart, cmp;
    if (pos_dis >= 0) { lstart = lmo + pos_dis; rstart = rmo; cmp = imin(lml - pos_dis, rml); }
    else              { lstart = lmo; rstart = rmo - pos_dis; cmp = imin(lml, rml + pos_dis); }
    uint8_t *lseq = seq_of(c, p->left), *rseq = seq_of(c, p->right);
    uint8_t *lq = qual_of(c, p->left), *rq = qual_of(c, p->right);
    assign_scores(o, lq, 0, imin(lk->l_qseq, lstart), p->lscore);
    assign_scores(o, lq, imax(0, lstart + cmp), lk->l_qseq, p->lscore);
    assign_scores(o, rq, 0, imin(rk->l_qseq, rstart), p->rscore);
    assign_scores(o, rq, imax(0, rstart + cmp), rk->l_qseq, p->rscore);
    for (int i = 0; i < cmp; i++) {
        int l = lstart + i, r = rstart + i;
        uint8_t ql = lq[l], qr = rq[r];
        if (nib(lseq, l) == nib(rseq, r)) {                 /* matched: score + 4, :148-154 */
            uint8_t q = (uint8_t)((ql + qr) / 2);
            signed char s = (signed char)(qual2score(o, q) + 4);
            p->lscore[l] = s; p->rscore[r] = s;
        } else {                                            /* mismatched: quals rewritten in place, :155-168 */
            lq[l] = (uint8_t)imax(0, (int)ql - (int)qr);
            rq[r] = (uint8_t)imax(0, (int)qr - (int)ql);
            if (ql >= qr) { p->lscore[l] = (signed char)(qual2score(o, (uint8_t)(ql - qr)) - 3); p->rscore[r] = 0; }
            else          { p->lscore[l] = 0; p->rscore[r] = (signed char)(qual2score(o, (uint8_t)(qr - ql)) - 3); }
        }
    }
}
static signed char *pair_left_score(Ctx *c, OPair *p)  { if (!p->lscore) pair_compute_score(c, p); return p->lscore; }  /* pair.cpp:174-179 */
static signed char *pair_right_score(Ctx *c, OPair *p) { if (!p->rscore) pair_compute_score(c, p); return p->rscore; }  /* pair.cpp:181-186 */

/* Pair::writeSscsDcsTag / writeSscsDcsTagBam                                                  pair.cpp:43-68.
 * The aux payload is 1 byte taken from the address of an unsigned short => low byte (quirk Q8). */
static void pair_write_tag(Ctx *c, OPair *p) {
    int fr = imin(p->merge_reads, 65535) & 0xFF;
    int rr = imin(p->reverse_merge_reads, 65535) & 0xFF;
    uint32_t rd[2] = {p->left, p->right};
    for (int k = 0; k < 2; k++) {
        if (rd[k] == NONE) continue;
        c->res->fr[rd[k]] = (int16_t)fr;
        if (p->is_duplex) c->res->rr[rd[k]] = (int16_t)rr;
    }
}

static void pair_free(OPair *p) { if (!p) return; free(p->lscore); free(p->rscore); free(p); }

/* ------------------------------------------------------------------------------------------------- Group */

/* BamUtil::copyQName                                                                     bamutil.cpp:338-366 */
static void copy_qname(Ctx *c, uint32_t from, uint32_t to) {
    if (lqname_of(c, to) < lqname_of(c, from)) { fail(c, GCE_ERR_QNAME_SHORT, "copyQName ERROR: desitination qname is shorter"); return; }
    c->res->qname_src[to] = c->res->qname_src[from];
}

/* Group::makeConsensus                                                                     group.cpp:320-579 */
static int make_consensus(Ctx *c, const uint32_t *reads, int n_reads, uint32_t out, signed char **scores, int is_left) {
    const gce_params *o = c->prm;
    const gce_core *ok = core_of(c, out);
    int diff = 0, mismatch_inc = 0;
    int seqbytes = (ok->l_qseq + 1) >> 1, qualbytes = ok->l_qseq;
    uint8_t *outdata = seq_of(c, out), *outqual = qual_of(c, out);
    uint8_t *seq_bak = (uint8_t *)malloc((size_t)seqbytes + 1), *qual_bak = (uint8_t *)malloc((size_t)qualbytes + 1);
    memcpy(seq_bak, outdata, (size_t)seqbytes);
    memcpy(qual_bak, outqual, (size_t)qualbytes);

    int *len_diff = (int *)malloc(sizeof(int) * (size_t)n_reads);
    for (int r = 0; r < n_reads; r++) {                     /* :339-348 */
        const gce_core *rk = core_of(c, reads[r]);
        int d = rk->l_qseq - ok->l_qseq;
        if (d != 0 && rk->pos == ok->pos &&
            orc_is_part_of(cigar_of(c, out), ok->n_cigar, cigar_of(c, reads[r]), rk->n_cigar, 1)) d = 0;   /* the "WAR" */
        len_diff[r] = d;
    }
    int len = ok->l_qseq;
    if (ok->n_cigar == 0)                                   /* :354-360 */
        for (int r = 0; r < n_reads; r++) if (core_of(c, reads[r])->l_qseq < len) len = core_of(c, reads[r])->l_qseq;

    const uint8_t *refdata = NULL; int64_t ref_n = 0;
    if (ok->isize != 0)                                     /* :362-367 */
        refdata = reference_data(c, ok->tid, ok->pos, (int64_t)orc_ref_offset(cigar_of(c, out), ok->n_cigar, len - 1) + 1, &ref_n);

    for (int i = 0; i < len; i++) {                         /* :369-526 */
        int counts[16] = {0}, base_scores[16] = {0}, quals[16] = {0};
        uint8_t top_quals[16] = {0};
        int total_score = 0;
        for (int r = 0; r < n_reads; r++) {
            int readpos = is_left ? i : i + len_diff[r];
            if (readpos < 0 || readpos >= core_of(c, reads[r])->l_qseq) continue;   /* out of bounds is UB in the reference; skipped by oracle AND engine */
            int base = nib(seq_of(c, reads[r]), readpos);
            uint8_t q = qual_of(c, reads[r])[readpos];
            counts[base]++;
            base_scores[base] += scores[r][readpos];
            total_score += scores[r][readpos];
            quals[base] += q;
            if (q > top_quals[base]) top_quals[base] = q;
        }
        int top_base = 0, top_score = -0x7FFFFFFF;          /* :394-402, `>=` makes the LATER bin win ties (Q6) */
        for (int bb = 0; bb < 16; bb++)
            if (base_scores[bb] > top_score || (base_scores[bb] == top_score && quals[bb] >= quals[top_base])) { top_score = base_scores[bb]; top_base = bb; }
        int top_num = counts[top_base];
        uint8_t top_qual = top_quals[top_base];
        int sec_base = 0, sec_score = -0x7FFFFFFF;          /* :406-416 */
        for (int bb = 0; bb < 16; bb++) {
            if (bb == top_base) continue;
            if (base_scores[bb] > sec_score || (base_scores[bb] == sec_score && quals[bb] >= quals[sec_base])) { sec_score = base_scores[bb]; sec_base = bb; }
        }
        int sec_num = counts[sec_base];
        int need_ref = 0;
        if (sec_num == 0) {                                 /* :421-428 */
            if (top_score >= o->base_score_req && top_qual >= o->moderate_quality) { outqual[i] = top_qual; continue; }
            need_ref = 1;
        }
        char refbase = 0;                                   /* :430-439 */
        if (refdata) {
            int refpos = orc_ref_offset(cigar_of(c, out), ok->n_cigar, i);
            if (refpos >= 0 && (int64_t)ok->pos + refpos < ref_n) refbase = orc_reference_base(refdata, (int64_t)ok->pos + refpos);
        }
        if (refbase != 'A' && refbase != 'T' && refbase != 'C' && refbase != 'G') refbase = 0;
        if (sec_num == 1) {                                 /* :442-457 */
            if (quals[sec_base] <= o->low_quality) { if (top_num < 2 && top_qual < o->high_quality) need_ref = 1; }
            else { if (top_num < 3 || top_qual < o->high_quality) need_ref = 1; }
        }
        if (sec_num > 1)                                    /* :460-464, double arithmetic (Q12) */
            if ((double)top_score < o->score_percent_req * total_score || top_qual < o->moderate_quality) need_ref = 1;
        if (top_score < o->base_score_req || top_qual <= o->low_quality) need_ref = 1;   /* :466-467 */

        if (need_ref && refbase != 0) {                     /* :470-501 */
            int ref4 = base2fourbits(refbase);
            signed char ref_base_qual = 0;                  /* `char refBaseQual` */
            for (int r = 0; r < n_reads; r++) {
                int readpos = is_left ? i : i + len_diff[r];
                if (readpos < 0 || readpos >= core_of(c, reads[r])->l_qseq) continue;
                int base = nib(seq_of(c, reads[r]), readpos);
                uint8_t q = qual_of(c, reads[r])[readpos];
                if (base == ref4) {
                    if ((int)q > (int)ref_base_qual) ref_base_qual = (signed char)q;
                    if (q >= o->high_quality) top_base = ref4;
                }
            }
            if (top_qual < o->moderate_quality) top_base = ref4;
            if (top_base == ref4) top_qual = (uint8_t)ref_base_qual;
        }
        int out_base = nib(outdata, i);                     /* :503-525 */
        if (out_base != top_base) {
            if (i & 1) outdata[i / 2] = (uint8_t)((outdata[i / 2] & 0xF0) | top_base);
            else       outdata[i / 2] = (uint8_t)((outdata[i / 2] & 0x0F) | (top_base << 4));
            diff++;
            if (refbase != 0) {
                int ref4 = base2fourbits(refbase);
                if (out_base == ref4) mismatch_inc++;
                else if (top_base == ref4) mismatch_inc--;
            }
        }
        outqual[i] = top_qual;
    }
    if (mismatch_inc != 0) {                                /* :528-573 */
        if (c->b->nm_type[out] == 0) fail(c, GCE_ERR_NM_MISSING, "NM tag missing while mismatchInc != 0 (reference dereferences NULL)");
        else {
            int new_nm = c->b->nm[out] + mismatch_inc;
            if (mismatch_inc > 5) {                         /* abnormal: restore (Q7) */
                memcpy(outdata, seq_bak, (size_t)seqbytes);
                memcpy(outqual, qual_bak, (size_t)qualbytes);
            } else if (c->b->nm_type[out] == 'C' && new_nm >= 0 && new_nm <= 255) c->res->nm_new[out] = new_nm;
        }
    }
    free(seq_bak); free(qual_bak); free(len_diff);
    return diff;
}

/* Group::consensusMergeBam                                                                 group.cpp:136-318.
 * `pairs` is the group's map<string,Pair*> in key (qname) order. Returns the template read or NONE. */
static uint32_t consensus_merge_bam(Ctx *c, OPair **pairs, int np, int is_left) {
    const gce_params *o = c->prm;
#define SIDE(p) (is_left ? (p)->left : (p)->right)
    if (np > o->skip_low_complexity_cluster_threshold) {    /* :142-175 */
        /* number of distinct CIGAR strings == number of distinct CIGAR word arrays */
        int distinct = 0; uint32_t first = NONE;
        uint32_t *seen = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np);
        for (int i = 0; i < np; i++) {
            uint32_t b = SIDE(pairs[i]);
            if (b == NONE) continue;
            if (first == NONE) first = b;
            int dup = 0;
            for (int j = 0; j < distinct && !dup; j++) {
                uint32_t s = seen[j];
                if (core_of(c, s)->n_cigar == core_of(c, b)->n_cigar &&
                    memcmp(cigar_of(c, s), cigar_of(c, b), 4u * core_of(c, b)->n_cigar) == 0) dup = 1;
            }
            if (!dup) seen[distinct++] = b;
        }
        free(seen);
        if ((double)distinct > np * 0.1 && first != NONE) {
            int n = core_of(c, first)->l_qseq, diff_neighbor = 0;
            for (int i = 0; i < n - 1; i++)
                if (fourbits2base(nib(seq_of(c, first), i)) != fourbits2base(nib(seq_of(c, first), i + 1))) diff_neighbor++;
            if ((double)diff_neighbor < n * 0.5) return NONE;
        }
    }
    int left_read_mode = is_left;                           /* :177-194 */
    if (!is_left) {
        int left_aligned = 1, last_pos = -1;
        for (int i = 0; i < np; i++) {
            if (pairs[i]->right == NONE) continue;
            int p = core_of(c, pairs[i]->right)->pos;
            if (last_pos >= 0 && p != last_pos) { left_aligned = 0; break; }
            last_pos = p;
        }
        if (left_aligned) left_read_mode = 1;
    }
    int *contained = (int *)calloc((size_t)np, sizeof(int));   /* :196-233 */
    for (int i = 0; i < np; i++) {
        uint32_t part = SIDE(pairs[i]);
        if (part == NONE) continue;
        int cb = 1;
        for (int j = 0; j < np; j++) {
            if (i == j) continue;
            uint32_t whole = SIDE(pairs[j]);
            if (whole == NONE) continue;
            if (!is_left && right_ref_pos(c, part) != right_ref_pos(c, whole)) continue;
            if (orc_is_part_of(cigar_of(c, part), core_of(c, part)->n_cigar, cigar_of(c, whole), core_of(c, whole)->n_cigar, left_read_mode)) cb++;
        }
        contained[i] = cb;
        if (np > o->skip_low_complexity_cluster_threshold && cb >= np / 2) break;
    }
    int best = -1, best_num = -1;                           /* :235-261 */
    for (int i = 0; i < np; i++) {
        if (contained[i] > best_num) { best_num = contained[i]; best = i; }
        else if (contained[i] == best_num && best >= 0) {
            int this_len = 0, cur_len = 0;
            if (SIDE(pairs[i]) != NONE) this_len = core_of(c, SIDE(pairs[i]))->l_qseq;
            if (SIDE(pairs[best]) != NONE) cur_len = core_of(c, SIDE(pairs[best]))->l_qseq;
            if (this_len < cur_len) { best_num = contained[i]; best = i; }
        }
    }
    free(contained);
    if ((double)best_num < np * 0.4 && np != 1) return NONE;   /* "no marjority", :264-266 */

    uint32_t out; signed char *out_score;                   /* :268-285 */
    if (is_left) { out = pairs[best]->left;  out_score = pair_left_score(c, pairs[best]);  pairs[best]->left = NONE; }
    else         { out = pairs[best]->right; out_score = pair_right_score(c, pairs[best]); pairs[best]->right = NONE; }
    if (out == NONE) return NONE;

    uint32_t *reads = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np);
    signed char **scores = (signed char **)malloc(sizeof(signed char *) * (size_t)np);
    int nr = 0;
    reads[nr] = out; scores[nr] = out_score; nr++;
    for (int j = 0; j < np; j++) {                          /* :293-313 */
        if (j == best) continue;
        uint32_t rd = SIDE(pairs[j]);
        signed char *sc = is_left ? pair_left_score(c, pairs[j]) : pair_right_score(c, pairs[j]);
        if (rd == NONE || sc == NULL) continue;
        if (orc_is_part_of(cigar_of(c, out), core_of(c, out)->n_cigar, cigar_of(c, rd), core_of(c, rd)->n_cigar, left_read_mode)) { reads[nr] = rd; scores[nr] = sc; nr++; }
    }
    make_consensus(c, reads, nr, out, scores, left_read_mode);   /* :315 */
    free(reads); free(scores);
    return out;
#undef SIDE
}

/* Group::consensusMerge                                                                     group.cpp:68-134 */
static OPair *consensus_merge(Ctx *c, OPair **pairs, int np, int cross_contig) {
    if (np == 1 && pairs[0]->right == NONE) { OPair *p = pairs[0]; pairs[0] = NULL; return p; }   /* :73-77 */
    uint32_t name_to_copy = NONE;                           /* :79-99 */
    if (cross_contig) {
        int cur_len = 0;
        for (int i = 0; i < np; i++) {
            uint32_t l = pairs[i]->left;
            if (l == NONE) continue;
            if (name_to_copy == NONE) { name_to_copy = l; cur_len = lqname_of(c, l); continue; }
            int ll = lqname_of(c, l);
            if (ll < cur_len || (ll == cur_len && strcmp(qname_of(c, l), qname_of(c, name_to_copy)) < 0)) { name_to_copy = l; cur_len = ll; }
        }
    }
    uint32_t left = consensus_merge_bam(c, pairs, np, 1);   /* :101-102 */
    uint32_t right = consensus_merge_bam(c, pairs, np, 0);
    OPair *p = (OPair *)calloc(1, sizeof(OPair));
    p->left = p->right = NONE;
    p->merge_reads = np;                                    /* :105 */
    if (cross_contig) { if (left != NONE && name_to_copy != NONE && name_to_copy != left) copy_qname(c, name_to_copy, left); }   /* :109-112 */
    else if (left != NONE && right != NONE) {               /* :114-123: std::string lengths == padded l_qname */
        if (lqname_of(c, left) <= lqname_of(c, right)) copy_qname(c, left, right);
        else copy_qname(c, right, left);
    }
    if (left != NONE) pair_set_left(c, p, left);
    if (right != NONE) pair_set_right(c, p, right);
    return p;
}

/* ----------------------------------------------------------------------------------------------- Cluster */

/* Cluster::duplexMergeBam                                                               cluster.cpp:199-244 */
static int duplex_merge_bam(Ctx *c, uint32_t b1, uint32_t b2) {
    int len1 = core_of(c, b1)->l_qseq, len2 = core_of(c, b2)->l_qseq;
    int diff = len1 > len2 ? len1 - len2 : len2 - len1;
    int len = imin(len1, len2);
    uint8_t *s1 = seq_of(c, b1), *s2 = seq_of(c, b2), *q1 = qual_of(c, b1), *q2 = qual_of(c, b2);
    for (int i = 0; i < len; i++) {
        if (s1[i / 2] == s2[i / 2]) { i++; continue; }      /* whole byte identical: skip both bases */
        char base1 = fourbits2base(nib(s1, i)), base2 = fourbits2base(nib(s2, i));
        if (base1 != base2) {
            diff++;
            q1[i] = 0; q2[i] = 0;
            if (i & 1) { s1[i / 2] = (uint8_t)((s1[i / 2] & 0xF0) | 15); s2[i / 2] = (uint8_t)((s2[i / 2] & 0xF0) | 15); }
            else       { s1[i / 2] = (uint8_t)((s1[i / 2] & 0x0F) | 0xF0); s2[i / 2] = (uint8_t)((s2[i / 2] & 0x0F) | 0xF0); }
        }
    }
    return diff;
}
/* Cluster::duplexMerge                                                                  cluster.cpp:190-197 */
static int duplex_merge(Ctx *c, OPair *p1, OPair *p2) {
    int diff = 0;
    if (p1->left != NONE && p2->left != NONE) diff += duplex_merge_bam(c, p1->left, p2->left);
    if (p1->right != NONE && p2->right != NONE) diff += duplex_merge_bam(c, p1->right, p2->right);
    return diff;
}

/* Gencore::outputPair                                                                  gencore.cpp:145-160 */
static void output_pair(Ctx *c, OPair *p) {
    stats_add_molecule(&c->res->post, 1, p->left != NONE && p->right != NONE);
    if (p->left != NONE)  { c->res->out_flag[p->left] = 1;  c->res->mate[p->left] = p->right; }
    if (p->right != NONE) { c->res->out_flag[p->right] = 1; c->res->mate[p->right] = p->left; }
}

typedef struct { Ctx *c; } SortCtx;
static Ctx *g_sort_ctx;                                     /* qsort has no user pointer; the oracle is single-threaded like the reference */
static int cmp_read_qname(const void *a, const void *b) {
    uint32_t ra = *(const uint32_t *)a, rb = *(const uint32_t *)b;
    int r = strcmp(g_sort_ctx->b->qname + g_sort_ctx->b->qname_off[ra], g_sort_ctx->b->qname + g_sort_ctx->b->qname_off[rb]);
    if (r) return r;
    return ra < rb ? -1 : ra > rb;                          /* arrival order == input order inside a cluster */
}
static int cmp_slice(const char *a, int la, const char *b, int lb) {   /* std::string operator< */
    int m = la < lb ? la : lb;
    int r = memcmp(a, b, (size_t)m);
    if (r) return r;
    return la - lb;
}

static OPair **g_sort_pairs;
static int cmp_pair_umi(const void *a, const void *b) {
    const OPair *x = g_sort_pairs[*(const int *)a], *y = g_sort_pairs[*(const int *)b];
    int r = cmp_slice(x->umi, x->umi_len, y->umi, y->umi_len);
    if (r) return r;
    return *(const int *)a - *(const int *)b;
}

/* Cluster::addRead (cluster.cpp:260-273) for all reads of a cluster, then Cluster::clusterByUMI (cluster.cpp:55-188) */
static void cluster_by_umi(Ctx *c, OCluster *cl, int umi_diff_threshold, int cross_contig) {
    const gce_params *o = c->prm;
    /* --- addRead: map<string,Pair*> keyed by qname; first read seen = left, later ones replace right --- */
    g_sort_ctx = c;
    qsort(cl->reads, (size_t)cl->n, sizeof(uint32_t), cmp_read_qname);
    OPair **pairs = (OPair **)malloc(sizeof(OPair *) * (size_t)cl->n);
    int np = 0;
    for (int i = 0; i < cl->n;) {
        int j = i + 1;
        while (j < cl->n && strcmp(qname_of(c, cl->reads[i]), qname_of(c, cl->reads[j])) == 0) j++;
        OPair *p = (OPair *)calloc(1, sizeof(OPair));
        p->left = p->right = NONE; p->merge_reads = 1;
        pair_set_left(c, p, cl->reads[i]);
        for (int k = i + 1; k < j; k++) pair_set_right(c, p, cl->reads[k]);
        pairs[np++] = p;
        i = j;
    }
    c->res->n_pairs += np;
    if (c->failed) { for (int i = 0; i < np; i++) pair_free(pairs[i]); free(pairs); return; }

    /* --- greedy UMI grouping, cluster.cpp:57-100 ---
     * map<string,int> umiCount is restated as: the distinct UMIs in std::string order (`cls`), one count each. */
    int has_umi = 0;
    for (int i = 0; i < np; i++) if (pairs[i]->umi_len) has_umi = 1;
    int *order = (int *)malloc(sizeof(int) * (size_t)np);
    for (int i = 0; i < np; i++) order[i] = i;
    g_sort_pairs = pairs;
    qsort(order, (size_t)np, sizeof(int), cmp_pair_umi);
    int *cls_of = (int *)malloc(sizeof(int) * (size_t)np);      /* pair -> class */
    int *cls_count = (int *)calloc((size_t)np, sizeof(int));    /* umiCount[umi] */
    int *cls_rep = (int *)malloc(sizeof(int) * (size_t)np);     /* a pair carrying that UMI */
    int n_cls = 0;
    for (int k = 0; k < np; k++) {
        int i = order[k];
        if (k == 0 || cmp_slice(pairs[i]->umi, pairs[i]->umi_len, pairs[order[k - 1]]->umi, pairs[order[k - 1]]->umi_len) != 0) { cls_rep[n_cls] = i; n_cls++; }
        cls_of[i] = n_cls - 1;
        cls_count[n_cls - 1]++;
    }
    int *group_of = (int *)malloc(sizeof(int) * (size_t)np);
    for (int i = 0; i < np; i++) group_of[i] = -1;
    int n_groups = 0, remaining = np;
    while (remaining > 0) {                                 /* while(mPairs.size()>0), :66 */
        int top = -1, top_count = 0;                        /* :68-76: strict > over the map in key order */
        for (int k = 0; k < n_cls; k++) if (cls_count[k] > top_count) { top_count = cls_count[k]; top = k; }
        const char *tu = ""; int tl = 0;
        if (top >= 0) { tu = pairs[cls_rep[top]]->umi; tl = pairs[cls_rep[top]]->umi_len; }
        for (int i = 0; i < np; i++)                        /* :83-95, pairs visited in qname order */
            if (group_of[i] < 0 && orc_umi_diff(pairs[i]->umi, pairs[i]->umi_len, tu, tl) <= umi_diff_threshold) {
                group_of[i] = n_groups; remaining--;
                cls_count[cls_of[i]] = 0;                   /* umiCount[umi] = 0, :91 */
            }
        if (top >= 0) cls_count[top] = 0;                   /* :99 */
        n_groups++;
    }
    free(order); free(cls_of); free(cls_count); free(cls_rep);
    stats_add_cluster(&c->res->pre, n_groups > 1);          /* :102 */
    c->res->n_clusters++;
    c->res->n_groups += n_groups;

    /* --- consensus per group, cluster.cpp:107-114 --- */
    OPair **single = (OPair **)malloc(sizeof(OPair *) * (size_t)n_groups);
    OPair **gp = (OPair **)malloc(sizeof(OPair *) * (size_t)np);
    for (int g = 0; g < n_groups; g++) {
        int gn = 0;
        for (int i = 0; i < np; i++) if (group_of[i] == g) gp[gn++] = pairs[i];   /* qname order preserved */
        single[g] = consensus_merge(c, gp, gn, cross_contig);
        for (int i = 0; i < gn; i++) if (gp[i] && gp[i] != single[g]) pair_free(gp[i]);
    }
    free(gp); free(group_of); free(pairs);

    /* --- duplex merge / filter, cluster.cpp:116-188 --- */
    int n_result = 0, ns = n_groups;
    if (has_umi && !o->disable_duplex) {
        while (ns > 0) {
            OPair *p1 = single[--ns];
            int found = 0;
            for (int i = 0; i < ns; i++) {
                OPair *p2 = single[i];
                if (!orc_is_duplex(p1->umi, p1->umi_len, p2->umi, p2->umi_len)) continue;
                found = 1;
                int diff = duplex_merge(c, p1, p2);
                stats_add_molecule(&c->res->pre, (unsigned)(p1->merge_reads + p2->merge_reads), p1->left != NONE && p1->right != NONE);
                if (diff <= o->duplex_mismatch_threshold && p1->merge_reads + p2->merge_reads >= o->cluster_size_req) {
                    p1->is_duplex = 1; p1->reverse_merge_reads = p2->merge_reads;
                    pair_write_tag(c, p1);
                    c->res->post.dcs++;
                    output_pair(c, p1); n_result++;
                }
                pair_free(p1);
                memmove(&single[i], &single[i + 1], sizeof(OPair *) * (size_t)(ns - i - 1));
                ns--;
                pair_free(p2);
                break;
            }
            if (!found) {
                stats_add_molecule(&c->res->pre, (unsigned)p1->merge_reads, p1->left != NONE && p1->right != NONE);
                if (!o->duplex_only && p1->merge_reads >= o->cluster_size_req) {
                    pair_write_tag(c, p1);
                    c->res->post.sscs++;
                    output_pair(c, p1); n_result++;
                }
                pair_free(p1);
            }
        }
    } else {
        for (int i = 0; i < ns; i++) {
            OPair *p = single[i];
            stats_add_molecule(&c->res->pre, (unsigned)p->merge_reads, p->left != NONE && p->right != NONE);
            if (!o->duplex_only && p->merge_reads >= o->cluster_size_req) {
                pair_write_tag(c, p);
                c->res->post.sscs++;
                output_pair(c, p); n_result++;
            }
            pair_free(p);
        }
    }
    if (n_result > 0) stats_add_cluster(&c->res->post, n_result > 1);   /* :185-187 */
    free(single);
}

/* ----------------------------------------------------------------------------------------- stream driver */

static uint64_t key_hash(int32_t tid, int32_t left, int64_t right) {
    uint64_t h = (uint64_t)(uint32_t)tid * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)left * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)right * 0x165667B19E3779F9ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
static void htab_rebuild(Ctx *c) {
    uint64_t need = 64;
    while (need < (uint64_t)(c->n_pending + 20000) * 2) need <<= 1;
    if (need - 1 != c->hmask || !c->htab) { free(c->htab); c->htab = (uint32_t *)malloc(sizeof(uint32_t) * need); c->hmask = need - 1; }
    memset(c->htab, 0, sizeof(uint32_t) * (c->hmask + 1));
    for (int64_t i = 0; i < c->n_pending; i++) {
        OCluster *cl = c->pending[i];
        uint64_t h = key_hash(cl->tid, cl->left, cl->right) & c->hmask;
        while (c->htab[h]) h = (h + 1) & c->hmask;
        c->htab[h] = (uint32_t)i + 1;
    }
}
/* Gencore::createCluster + lookup                                                    gencore.cpp:449-467,316 */
static OCluster *get_cluster(Ctx *c, int32_t tid, int32_t left, int64_t right) {
    if ((uint64_t)(c->n_pending + 1) * 2 > c->hmask + 1) htab_rebuild(c);
    uint64_t h = key_hash(tid, left, right) & c->hmask;
    while (c->htab[h]) {
        OCluster *cl = c->pending[c->htab[h] - 1];
        if (cl->tid == tid && cl->left == left && cl->right == right) return cl;
        h = (h + 1) & c->hmask;
    }
    if (c->n_pending == c->cap_pending) {
        c->cap_pending = c->cap_pending ? c->cap_pending * 2 : 1024;
        c->pending = (OCluster **)realloc(c->pending, sizeof(OCluster *) * (size_t)c->cap_pending);
    }
    OCluster *cl = (OCluster *)calloc(1, sizeof(OCluster));
    cl->tid = tid; cl->left = left; cl->right = right;
    c->pending[c->n_pending] = cl;
    c->htab[h] = (uint32_t)c->n_pending + 1;
    c->n_pending++;
    return cl;
}
static void cluster_free(OCluster *cl) { free(cl->reads); free(cl); }

static int cmp_cluster_key(const void *a, const void *b) {
    const OCluster *x = *(OCluster *const *)a, *y = *(OCluster *const *)b;
    if (x->tid != y->tid) return x->tid < y->tid ? -1 : 1;
    if (x->left != y->left) return x->left < y->left ? -1 : 1;
    if (x->right != y->right) return x->right < y->right ? -1 : 1;
    return 0;
}

/* the periodic flush walk of Gencore::addToProperCluster                             gencore.cpp:324-389.
 * The three nested map loops are flattened over the key-sorted pending set; the break rules are kept literally:
 *   - outer: stop at the first tid > current tid, or once `needBreak` was raised           (:334)
 *   - middle: on the current tid, the first left >= pos raises needBreak and stops          (:344-349)
 *   - inner: on the current tid, the first right >= pos ends this left's walk               (:352-354)  */
static void periodic_flush(Ctx *c, int32_t tid, int32_t pos) {
    qsort(c->pending, (size_t)c->n_pending, sizeof(OCluster *), cmp_cluster_key);
    int64_t keep = 0, i = 0;
    int need_break = 0;
    while (i < c->n_pending) {
        OCluster *cl = c->pending[i];
        if (cl->tid > tid || need_break) break;
        if (cl->tid == tid && cl->left >= pos) { need_break = 1; break; }
        /* walk the rights of this (tid,left) */
        int64_t j = i;
        int stopped = 0;
        while (j < c->n_pending && c->pending[j]->tid == cl->tid && c->pending[j]->left == cl->left) {
            OCluster *cr = c->pending[j];
            if (!stopped && cr->tid == tid && cr->right >= pos) stopped = 1;
            if (stopped) c->pending[keep++] = cr;
            else {
                cluster_by_umi(c, cr, c->prm->proper_umi_diff_threshold, cr->right < 0);   /* :355 */
                cluster_free(cr);
            }
            j++;
        }
        i = j;
    }
    while (i < c->n_pending) c->pending[keep++] = c->pending[i++];
    c->n_pending = keep;
    htab_rebuild(c);
}

/* Gencore::finishConsensus                                                            gencore.cpp:392-434 */
static void finish_consensus(Ctx *c, int umi_diff_threshold) {
    qsort(c->pending, (size_t)c->n_pending, sizeof(OCluster *), cmp_cluster_key);
    for (int64_t i = 0; i < c->n_pending; i++) {
        OCluster *cl = c->pending[i];
        /* tid < 0 || left < 0 clusters ("unmapped", :401-407) cannot exist: such reads never reach addToCluster */
        cluster_by_umi(c, cl, umi_diff_threshold, cl->right < 0);                          /* :409, threshold quirk Q1 */
        cluster_free(cl);
    }
    c->n_pending = 0;
    htab_rebuild(c);
}

/* Gencore::addToProperCluster                                                         gencore.cpp:295-390 */
static void add_to_proper_cluster(Ctx *c, uint32_t r) {
    const gce_core *k = core_of(c, r);
    int32_t tid = k->tid, left = k->pos;
    int64_t right;
    int64_t d = (int64_t)k->mpos - (int64_t)k->pos; if (d < 0) d = -d;
    if (k->mtid == k->tid && d < 100000) {                  /* :300-304 */
        if (k->isize < 0) left = k->mpos;
        int64_t a = k->isize; if (a < 0) a = -a;
        right = (int64_t)left + a - 1;
    } else {
        if (k->mtid < 0) { c->res->out_flag[r] = 2; return; }   /* :307-309: written as is, no cluster, no tick */
        int64_t tl = (tid < c->prm->n_targets && c->prm->target_len) ? (int64_t)c->prm->target_len[tid] : 0;
        right = -1 * tl * (int64_t)(k->mtid + 1) + (int64_t)k->mpos;   /* :311 */
    }
    OCluster *cl = get_cluster(c, tid, left, right);        /* :315-316 */
    if (cl->n == cl->cap) { cl->cap = cl->cap ? cl->cap * 2 : 4; cl->reads = (uint32_t *)realloc(cl->reads, sizeof(uint32_t) * (size_t)cl->cap); }
    cl->reads[cl->n++] = r;
    int period = c->prm->flush_period > 0 ? c->prm->flush_period : 10000;
    if (c->b->tick) {
        /* key-range shard: the global tick of this read comes with the batch, and the flush events of reads that live in
         * OTHER shards (they fire between two of this shard's reads) are replayed from the global event table first */
        const int64_t t = (int64_t)c->b->tick[r];
        while (c->next_ev < c->n_ev && (int64_t)(c->next_ev + 1) * period < t) { periodic_flush(c, c->ev_tid[c->next_ev], c->ev_pos[c->next_ev]); c->next_ev++; }
        c->tick = t;
        if (t % period == 0 && c->next_ev < c->n_ev && (int64_t)(c->next_ev + 1) * period == t) c->next_ev++;   /* this read IS event next_ev */
    } else
        c->tick++;                                          /* :319-322 */
    if (c->tick % period != 0) return;
    periodic_flush(c, tid, k->pos);
}

/* Gencore::consensus — the read loop                                                  gencore.cpp:205-279 */
int orc_run(const gce_params *prm, const orc_reference *ref, gce_batch *batch, orc_result *out) {
    return orc_run_shard(prm, ref, batch, 0, NULL, NULL, out);
}

/* The same over a key-range shard of a stream (gencore_amd/shard.py): batch->tick carries every read's global tick and
 * (ev_tid, ev_pos) are the reads on which the whole stream's periodic flushes fire (gencore.cpp:319-322). */
int orc_run_shard(const gce_params *prm, const orc_reference *ref, gce_batch *batch, int32_t n_events, const int32_t *ev_tid,
                  const int32_t *ev_pos, orc_result *out) {
    memset(out, 0, sizeof *out);
    int64_t n = batch->n_reads;
    out->n_reads = n;
    size_t nn = (size_t)(n > 0 ? n : 1);
    out->out_flag = (uint8_t *)calloc(nn, 1);
    out->qname_src = (uint32_t *)malloc(nn * sizeof(uint32_t));
    out->nm_new = (int32_t *)malloc(nn * sizeof(int32_t));
    out->fr = (int16_t *)malloc(nn * sizeof(int16_t));
    out->rr = (int16_t *)malloc(nn * sizeof(int16_t));
    out->mate = (uint32_t *)malloc(nn * sizeof(uint32_t));
    for (int64_t i = 0; i < n; i++) { out->qname_src[i] = (uint32_t)i; out->nm_new[i] = -1; out->fr[i] = -1; out->rr[i] = -1; out->mate[i] = NONE; }

    Ctx ctx; memset(&ctx, 0, sizeof ctx);
    ctx.prm = prm; ctx.ref = ref; ctx.b = batch; ctx.res = out;
    ctx.tick = prm->tick_offset;
    ctx.n_ev = batch->tick ? n_events : 0; ctx.ev_tid = ev_tid; ctx.ev_pos = ev_pos;
    htab_rebuild(&ctx);

    int last_tid = -1, last_pos = -1, out_set_cleared = 0, finished = 0;
    for (int64_t i = 0; i < n && !ctx.failed; i++) {
        const gce_core *k = &batch->core[i];
        stats_add_read(&out->pre, k, batch->nm_type[i] != 0, batch->nm[i]);     /* :222 */
        if (k->tid < last_tid || (k->tid == last_tid && k->pos < last_pos)) {   /* :233-241 */
            if (k->tid >= 0 && k->pos >= 0) { fail(&ctx, GCE_ERR_UNSORTED, "ERROR: the input is unsorted"); break; }
        }
        if (prm->max_contig > 0 && k->tid >= prm->max_contig) break;   /* :243-246 --quit_after_contig: counted and checked above, then the loop ends */
        last_tid = k->tid; last_pos = k->pos;
        if (k->tid < 0 || k->pos < 0) {                     /* :255-266: unmapped reads are dropped */
            if (!out_set_cleared) {
                if (!finished) { finished = 1; finish_consensus(&ctx, prm->unproper_umi_diff_threshold); }
                out_set_cleared = 1;
            }
            continue;
        }
        if (k->flag & (0x100 | 0x800)) continue;            /* BamUtil::isPrimary, :269-271 */
        add_to_proper_cluster(&ctx, (uint32_t)i);           /* :272 -> addToCluster :469-476 */
    }
    if (!ctx.failed) {
        /* :276-279.  In a coordinate-sharded run a flush event of a LATER slice (larger tid) would have drained
         * everything pending here through the periodic path, i.e. with -d instead of the end-of-file threshold. */
        while (!finished && ctx.next_ev < ctx.n_ev) { periodic_flush(&ctx, ctx.ev_tid[ctx.next_ev], ctx.ev_pos[ctx.next_ev]); ctx.next_ev++; }   /* later shards' events */
        if (!finished) { finished = 1; finish_consensus(&ctx, (prm->trailing_flush && !batch->tick) ? prm->proper_umi_diff_threshold : prm->unproper_umi_diff_threshold); }
    }
    /* clusters still pending after an earlier finish are never processed (released in ~Gencore, gencore.cpp:23) */
    for (int64_t i = 0; i < ctx.n_pending; i++) cluster_free(ctx.pending[i]);
    free(ctx.pending); free(ctx.htab);

    /* Gencore::writeBam -> mPostStats->addRead for every record that reaches the output (gencore.cpp:83-111) */
    if (!ctx.failed)
        for (int64_t i = 0; i < n; i++)
            if (out->out_flag[i]) {
                int nm = out->nm_new[i] >= 0 ? out->nm_new[i] : batch->nm[i];
                stats_add_read(&out->post, &batch->core[i], batch->nm_type[i] != 0, nm);
            }
    return out->status;
}

void orc_free_result(orc_result *r) {
    free(r->out_flag); free(r->qname_src); free(r->nm_new); free(r->fr); free(r->rr); free(r->mate);
    memset(r, 0, sizeof *r);
}

/* Stats::statDepth, the genome part                                                          stats.cpp:57-84 */
void orc_stat_depth(int64_t *depth, int64_t n_bins, int32_t step, int32_t start, int32_t len) {
    int end = start + len;
    int left_pos = start / step, right_pos = end / step;                 /* :66-67 (C division truncates toward zero) */
    if (right_pos >= n_bins || left_pos < 0) return;                     /* :69-70 */
    if (left_pos == right_pos) depth[left_pos] += len;                   /* :72-73 */
    else {
        depth[left_pos] += (left_pos + 1) * step - start;                /* :75-78 */
        depth[right_pos] += end - right_pos * step;
        for (int p = left_pos + 1; p < right_pos; p++) depth[p] += step; /* :80-82 */
    }
}
/* Bed::statDepth                                                                             bed.cpp:66-81 */
void orc_bed_depth(const int32_t *r_start, const int32_t *r_end, int64_t *r_count, int32_t n_regions, int32_t start, int32_t len) {
    int end = start + len;
    for (int p = 0; p < n_regions; p++) {
        if (r_end[p] < start) continue;                                  /* :73-74 */
        if (r_start[p] > end) break;                                     /* :75-76: assumes sorted regions */
        int l = (r_end[p] < end ? r_end[p] : end) - (r_start[p] > start ? r_start[p] : start);   /* :78 */
        r_count[p] += l;
    }
}

int orc_abi_version(void) { return GCE_ABI_VERSION; }
