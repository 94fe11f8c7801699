/*
 * gencore_amd.h — C-ABI of the MI355X consensus-read engine (libgencore_amd.so).
 *
 * This is the drop-in boundary for ONE path of OpenGene/gencore v0.17.2: everything between
 * Gencore::addToCluster(b) (reference src/gencore.cpp:361,558) and Gencore::outputPair(p)
 * (src/gencore.cpp:234), i.e. Cluster -> Group -> Pair -> consensus.  The reference has no FFI; the seam
 * it offers is the C++ call pair
 *     Cluster::addRead(bam1_t*)                                   src/cluster.h:23,  src/cluster.cpp:308
 *     vector<Pair*> Cluster::clusterByUMI(thr, pre, post, cross)  src/cluster.h:26,  src/cluster.cpp:103
 * driven by Gencore::addToProperCluster (src/gencore.cpp:384-479) and Gencore::finishConsensus
 * (src/gencore.cpp:481-523).  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, POD structs, pointers + sizes; no C++/torch types.
 *   - every function returns 0 (GCE_OK) or a negative gce_status; it NEVER calls exit().  The host adapter
 *     maps the codes back to the reference's messages + exit(-1)  (src/util.h:250 error_exit).
 *   - one engine per GPU per host thread; no global state inside the library.
 *   - the engine keeps the whole submitted stream resident in HBM (288 GB per MI355X) and processes it in
 *     one pass at gce_process(): the reference's 10,000-read flush cadence (src/gencore.cpp:408-411) is
 *     reproduced exactly as a per-cluster attribute, not as a host-side loop.
 */
#ifndef GENCORE_AMD_H
#define GENCORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCE_ABI_VERSION 1
#define GCE_NONE 0xFFFFFFFFu           /* "no read" marker in uint32 index arrays */
#define GCE_MAX_SUPPORTING_READS 100   /* src/stats.h:15 MAX_SUPPORTING_READS */

typedef struct gce_engine gce_engine;   /* opaque */

typedef enum gce_status {
    GCE_OK = 0,
    GCE_ERR_INVALID = -1,          /* bad argument / bad call order */
    GCE_ERR_NO_DEVICE = -2,        /* no HIP device: the product path has NO CPU fallback */
    GCE_ERR_HIP = -3,              /* HIP runtime failure, see gce_last_error() */
    GCE_ERR_OOM = -4,
    /* fatal conditions of the reference path, reported instead of exit(-1): */
    GCE_ERR_UNSORTED = -10,        /* src/gencore.cpp:322-329 "the input is unsorted" */
    GCE_ERR_UMI_MISMATCH = -11,    /* src/pair.cpp:605-616 "The UMI of a read pair should be identical" */
    GCE_ERR_NM_MISSING = -12,      /* src/group.cpp:581-584: NM dereferenced although absent (segfault in the reference) */
    GCE_ERR_UMI_PARSE = -13,       /* src/bamutil.cpp:86-102: substr(start) with start > length throws in the reference */
    GCE_ERR_QNAME_SHORT = -14      /* src/bamutil.cpp:383-386 "copyQName ERROR: desitination qname is shorter" */
} gce_status;

/* One read's fixed-size fields.  Byte-for-byte the BAM alignment core block (SAMv1 section 4.2), i.e. what
 * htslib's bam1_core_t is decoded from; 32 bytes, the "packed key record" of the clustering scan. */
typedef struct gce_core {
    int32_t  tid;       /* bam1_core_t.tid   (refID)                                    */
    int32_t  pos;       /* bam1_core_t.pos   0-based leftmost                           */
    uint8_t  l_qname;   /* strlen(qname)+1 as stored in BAM; the engine applies htslib's in-memory
                           padding to a multiple of 4 (l_extranul) wherever the reference compares
                           name lengths (src/group.cpp:143,167; src/bamutil.cpp:381-389)            */
    uint8_t  mapq;
    uint16_t bin;
    uint16_t n_cigar;
    uint16_t flag;
    int32_t  l_qseq;
    int32_t  mtid;
    int32_t  mpos;
    int32_t  isize;
} gce_core;

/* Mirror of the Options fields the path reads (src/options.h:37-60, defaults src/options.cpp:4-40). */
typedef struct gce_params {
    int32_t abi_version;                    /* GCE_ABI_VERSION */
    int32_t device;                         /* HIP device ordinal */
    int32_t proper_umi_diff_threshold;      /* -d  properReadsUmiDiffThreshold      = 1   */
    int32_t unproper_umi_diff_threshold;    /*     unproperReadsUmiDiffThreshold    = 0   */
    int32_t duplex_mismatch_threshold;      /* -D  duplexMismatchThreshold          = 2   */
    int32_t cluster_size_req;               /* -s  clusterSizeReq                   = 1   */
    int32_t base_score_req;                 /* -c  baseScoreReq                     = 6   */
    int32_t high_quality;                   /* --high_qual                          = 30  */
    int32_t moderate_quality;               /* --moderate_qual                      = 20  */
    int32_t low_quality;                    /* --low_qual                           = 15  */
    int32_t score_high;                     /* scoreOfNotOverlappedHighQual         = 8   */
    int32_t score_moderate;                 /* scoreOfNotOverlappedModerateQual     = 6   */
    int32_t score_low;                      /* scoreOfNotOverlappedLowQual          = 4   */
    int32_t score_bad;                      /* scoreOfNotOverlappedBadQual          = 2   */
    int32_t skip_low_complexity_cluster_threshold;  /*                              = 1000 */
    int32_t duplex_only;                    /* -x */
    int32_t disable_duplex;                 /* --no_duplex */
    int32_t flush_period;                   /* the literal 10000 of src/gencore.cpp:410 */
    double  score_percent_req;              /* -a  scorePercentReq                  = 0.8 */
    char    umi_prefix[32];                 /* -u, already resolved ("auto" -> gce_detect_umi_prefix) */
    /* contig lengths of the BAM header: needed by the cross-contig key, src/gencore.cpp:400 */
    int32_t n_targets;
    const uint32_t *target_len;             /* [n_targets], copied by gce_create */
    /* Stream context for coordinate-sharded (multi-GPU) runs: this engine sees a contiguous slice of the
     * globally sorted stream.  tick_offset = number of clustered reads before the slice (the reference's
     * static `tick`, src/gencore.cpp:408); trailing_flush != 0 if a flush event occurs after the slice, so
     * clusters still pending at the end of the slice get -d instead of the end-of-file threshold (quirk Q1). */
    int64_t tick_offset;
    int32_t trailing_flush;
    int32_t reserved;
} gce_params;

/* A batch of reads in INPUT ORDER (coordinate-sorted), struct-of-arrays.  Offsets are start offsets, lengths
 * come from gce_core (l_qname, n_cigar, l_qseq).  seq is BAM 4-bit packed (high nibble = even base,
 * src/bamutil.cpp:173-186), qual is raw Phred.  seq/qual are MUTATED IN PLACE exactly where the reference
 * mutates its bam1_t records (src/pair.cpp:562-563, src/group.cpp:560-574,604-605, src/cluster.cpp:275-288).
 * Device buffers (gce_submit_device): `core` must be 16-byte aligned and every blob readable 16 bytes past its end
 * (vector loads of the last read); gce_submit pads its own copies.  Reads longer than 65535 bases are rejected. */
typedef struct gce_batch {
    int64_t         n_reads;
    const gce_core *core;        /* [n] */
    const uint64_t *qname_off;   /* [n] byte offset into qname; names are NUL-terminated */
    const char     *qname;
    const uint64_t *cigar_off;   /* [n] offset in 32-bit words into cigar */
    const uint32_t *cigar;       /* BAM encoding len<<4|op */
    const uint64_t *seq_off;     /* [n] byte offset into seq */
    uint8_t        *seq;
    const uint64_t *qual_off;    /* [n] byte offset into qual */
    uint8_t        *qual;
    const int32_t  *nm;          /* [n] value of the NM aux tag (bam_aux2i), ignored when nm_type==0 */
    const uint8_t  *nm_type;     /* [n] BAM aux type byte of NM ('C','c','S','s','I','i'), 0 = tag absent */
    const uint64_t *mi_off;      /* optional: [n] offset of the MI:Z string, UINT64_MAX = read has no MI tag */
    const char     *mi;          /* optional: NUL-terminated MI:Z strings (src/bamutil.cpp:63-78); NULL if unused */
    size_t qname_bytes, cigar_words, seq_bytes, qual_bytes, mi_bytes;   /* total sizes of the blobs */
} gce_batch;

/* Additive QC counters touched on the path (src/stats.h:47-65; call sites src/gencore.cpp:199,235,311 and
 * src/cluster.cpp:150,184,190,206,210,221,225,233).  All int64, all additive => one RCCL all-reduce(sum). */
typedef struct gce_stats {
    int64_t reads;                     /* mRead                */
    int64_t bases;                     /* mBase                */
    int64_t reads_unmapped;            /* mReadUnmapped        */
    int64_t bases_unmapped;            /* mBaseUnmapped        */
    int64_t base_mismatches;           /* mBaseMismatches      */
    int64_t reads_with_mismatches;     /* mReadWithMismatches  */
    int64_t clusters;                  /* mCluster             */
    int64_t multi_molecule_clusters;   /* mMultiMoleculeCluster*/
    int64_t molecules;                 /* mMolecule            */
    int64_t molecules_se;              /* mMoleculeSE          */
    int64_t molecules_pe;              /* mMoleculePE          */
    int64_t sscs;                      /* mSSCSNum             */
    int64_t dcs;                       /* mDCSNum              */
    int64_t uncounted_supporting_reads;/* uncountedSupportingReads */
    int64_t supporting_hist[GCE_MAX_SUPPORTING_READS];  /* mSupportingHistgram */
} gce_stats;
#define GCE_STATS_WORDS (14 + GCE_MAX_SUPPORTING_READS)

/* Per-read result table (host pointers, owned by the engine until the next gce_process/gce_destroy).
 * An output record is "input read i, with its seq/qual as left in the mutated buffers, its qname replaced by
 * the qname of read qname_src[i], NM patched to nm_new[i] if >= 0, and FR/RR aux bytes appended if >= 0". */
typedef struct gce_result {
    int64_t         n_reads;
    const uint8_t  *out_flag;    /* [n] 0 = not emitted; 1 = emitted by outputPair (src/gencore.cpp:234);
                                        2 = mate-unmapped pass-through (src/gencore.cpp:396-398) */
    const uint32_t *qname_src;   /* [n] BamUtil::copyQName source (src/bamutil.cpp:378), == i if unchanged */
    const int32_t  *nm_new;      /* [n] -1 = NM untouched, else the byte written at src/group.cpp:619 */
    const int16_t  *fr;          /* [n] -1 = no tag, else the FR:C byte (src/pair.cpp:462-463, low byte, Q8) */
    const int16_t  *rr;          /* [n] -1 = no tag, else the RR:C byte (src/pair.cpp:467-468) */
    const uint32_t *mate;        /* [n] the other record of the same output Pair, or GCE_NONE */
    const uint8_t  *seq;         /* mutated copies of the submitted blobs (same offsets) */
    const uint8_t  *qual;
    int64_t         n_out;       /* number of reads with out_flag != 0 */
    const uint32_t *out_index;   /* [n_out] their indices, ascending */
    gce_stats       pre;         /* mPreStats  deltas */
    gce_stats       post;        /* mPostStats deltas */
} gce_result;

/* Kernel timing of the last gce_process(), HIP events on the engine's stream. */
typedef struct gce_timing {
    double total_ms;             /* first kernel start -> last kernel end */
    double prescan_ms;           /* read classification + tick scan  */
    double cluster_ms;           /* clustering scan (key + hash partition), the roofline kernel  */
    double csr_ms;               /* bucket offsets + scatter */
    double pairing_ms;           /* mate pairing + UMI grouping per cluster */
    double score_ms;             /* Pair::computeScore (mate-overlap patches) */
    double consensus_ms;         /* template pick + column vote (fast + generic kernels) */
    double finish_ms;            /* duplex merge, filter, tags, stats */
    int64_t n_clusters, n_groups, n_pairs;
} gce_timing;

/* Fill *p with the reference defaults (src/options.cpp:4-40). */
void gce_params_default(gce_params *p);

/* "auto" UMI prefix detection on the first read's qname, src/gencore.cpp:296-309.  Writes "", "umi" or "UMI". */
void gce_detect_umi_prefix(const char *first_qname, char out_prefix[32]);

/* Replaces: Gencore::Gencore(Options*) + Cluster(Options*) construction, src/gencore.cpp:7-19. */
int gce_create(const gce_params *params, gce_engine **out);
void gce_destroy(gce_engine *e);

/* Replaces: Reference::instance(opt)->getData(tid, ...) (src/reference.cpp:33-70) as the source of the staged
 * reference.  `nibbles` is one contig in FastaReader's own 4-bit code (A=1,T=2,C=3,G=4, other=0, LOW nibble =
 * even position; src/fastareader.cpp:106-128,139-152), n_bases its length.  Contigs never set behave like
 * "contig not found in the reference" (getData returns NULL). */
int gce_set_reference(gce_engine *e, int32_t tid, const uint8_t *nibbles, int64_t n_bases);
/* Convenience: pack an upper-cased ASCII contig into that code (FastaReader::to4bits, src/fastareader.cpp:139). */
void gce_pack_reference(const char *bases, int64_t n_bases, uint8_t *nibbles_out);

/* Replaces: the per-read loop body Gencore::addToCluster(b) (src/gencore.cpp:361,558-566) for a whole batch.
 * Host buffers are copied to HBM; the caller keeps ownership.  May be called repeatedly; batches are
 * concatenated in call order (they must continue the same sorted stream). */
int gce_submit(gce_engine *e, const gce_batch *batch);
/* Same, but every pointer in *batch is a DEVICE pointer that stays valid until gce_process() returns; seq/qual
 * are mutated in place in the caller's HBM buffers (zero-copy path used by bench.py and by a GPU BAM decoder). */
int gce_submit_device(gce_engine *e, const gce_batch *batch);

/* Replaces: every clusterByUMI call of the stream (src/gencore.cpp:444 periodic, :498 end of file) plus
 * outputPair bookkeeping.  Runs the whole HIP pipeline over everything submitted since the last process. */
int gce_process(gce_engine *e);

/* Replaces: draining csPairs into Gencore::outputPair (src/gencore.cpp:445-449,499-503).  Copies the result
 * table to host memory owned by the engine. */
int gce_drain(gce_engine *e, gce_result *out);
/* Device-side view of the same table (device pointers; seq/qual are the mutated device blobs). */
int gce_result_device(gce_engine *e, gce_result *out);

int gce_get_timing(gce_engine *e, gce_timing *out);
/* Drop all submitted reads/results but keep params, reference and allocations (for repeated bench steps). */
int gce_reset(gce_engine *e);

const char *gce_last_error(const gce_engine *e);   /* human-readable detail of the last failure */
const char *gce_status_message(int status);        /* the reference's message for a status code */
int gce_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GENCORE_AMD_H */
