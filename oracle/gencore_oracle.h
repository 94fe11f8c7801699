/*
 * gencore_oracle.h — CPU oracle for the Cluster -> Group -> consensus path of OpenGene/gencore v0.17.2.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gencore_amd/ (the product) may include, link or call this.
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY PINNING STATUS
 *   pinned   : orc_get_umi, orc_umi_diff, orc_is_duplex — checked against every known-answer vector the
 *              reference's own `gencore test` holds (src/bamutil.cpp:385-423, src/cluster.cpp:275-288).
 *   UNPINNED : consensus bases/quals, NM, FR/RR, Stats.  The reference ships no golden vector for them
 *               and cannot be built in this image: every .cpp includes htslib/sam.h and links
 *              -lhts (Makefile:17), htslib is not installed, and building against stand-in headers is not
 *              allowed.  For these outputs this file is a line-by-line restatement (each function cites the
 *              reference file:line it follows) — "parity unpinned" until the reference can be run.
 */
#ifndef GENCORE_ORACLE_H
#define GENCORE_ORACLE_H

#include "../include/gencore_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Reference::getData backing store (src/reference.cpp:33-70): per BAM tid one contig in FastaReader's 4-bit code
 * (src/fastareader.cpp:139-152) or NULL when the contig is missing from the FASTA. */
typedef struct orc_reference {
    int32_t         n_contigs;
    const uint8_t **data;      /* [n_contigs] */
    const int64_t  *n_bases;   /* [n_contigs] */
} orc_reference;

typedef struct orc_result {
    int64_t   n_reads;
    uint8_t  *out_flag;        /* same meaning as gce_result */
    uint32_t *qname_src;
    int32_t  *nm_new;
    int16_t  *fr;
    int16_t  *rr;
    uint32_t *mate;
    gce_stats pre, post;
    int64_t   n_clusters, n_groups, n_pairs;
    int       status;          /* gce_status */
    char      message[256];
} orc_result;

/* Run the whole path over one coordinate-sorted stream.  batch->seq / batch->qual are mutated in place. */
int  orc_run(const gce_params *prm, const orc_reference *ref, gce_batch *batch, orc_result *out);
int  orc_run_shard(const gce_params *prm, const orc_reference *ref, gce_batch *batch, int32_t n_events, const int32_t *ev_tid,
                   const int32_t *ev_pos, orc_result *out);   /* key-range shard: batch->tick + the stream's flush events */
void orc_free_result(orc_result *r);

/* unit-level entry points (known-answer tests) */
int  orc_get_umi(const char *name, const char *prefix, char *out, int cap);  /* returns length, -1 = reference throws */
int  orc_umi_diff(const char *a, int la, const char *b, int lb);
int  orc_is_duplex(const char *a, int la, const char *b, int lb);
int  orc_is_part_of(const uint32_t *part, int n_part, const uint32_t *whole, int n_whole, int is_left);
int  orc_ref_offset(const uint32_t *cigar, int n_cigar, int bampos);
void orc_m_offset_len(const uint32_t *cigar, int n_cigar, int *m_off, int *m_len);
int  orc_cigar_rlen(const uint32_t *cigar, int n_cigar);
void orc_pack_reference(const char *bases, int64_t n, uint8_t *out);
char orc_reference_base(const uint8_t *data, int64_t pos);
/* one group side in isolation: reads[0] is the template.  Used by kernel-level parity tests. */
/* Stats::statDepth + Bed::statDepth (src/stats.cpp:57-84, src/bed.cpp:66-81) for one read: depth[] is the contig's bin array
 * (1 + target_len / step entries), regions are the contig's BED regions in FILE order. */
void orc_stat_depth(int64_t *depth, int64_t n_bins, int32_t step, int32_t start, int32_t len);
void orc_bed_depth(const int32_t *r_start, const int32_t *r_end, int64_t *r_count, int32_t n_regions, int32_t start, int32_t len);
int  orc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
