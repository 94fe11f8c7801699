/*
 * gencore_amd.h — C-ABI of the MI355X consensus-read engine (libgencore_amd.so).
 *
 * This is the drop-in boundary for ONE path of OpenGene/gencore v0.17.2: everything between
 * Gencore::addToCluster(b) (call src/gencore.cpp:272, definition :469) and the output set that
 * Gencore::outputPair(p) (src/gencore.cpp:145) feeds, i.e. Cluster -> Group -> Pair -> consensus -> output order.
 * The reference has no FFI; the seam it offers is the C++ call pair
 *     Cluster::addRead(bam1_t*)                                   src/cluster.h:23,  src/cluster.cpp:260
 *     vector<Pair*> Cluster::clusterByUMI(thr, pre, post, cross)  src/cluster.h:26,  src/cluster.cpp:55
 * driven by Gencore::addToProperCluster (src/gencore.cpp:295-390) and Gencore::finishConsensus
 * (src/gencore.cpp:392-434).  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, POD structs, pointers + sizes; no C++/torch types.
 *   - every function returns 0 (GCE_OK) or a negative gce_status; it NEVER calls exit().  The host adapter
 *     maps the codes back to the reference's messages + exit(-1)  (src/util.h:250 error_exit).
 *   - one engine per GPU per host thread; no global state inside the library.
 *   - the engine keeps the whole submitted stream resident in HBM (288 GB per MI355X) and processes it in
 *     one pass at gce_process(): the reference's 10,000-read flush cadence (src/gencore.cpp:319-322) is
 *     reproduced exactly as a per-cluster attribute, not as a host-side loop.
 */
#ifndef GENCORE_AMD_H
#define GENCORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCE_ABI_VERSION 3              /* (struct layouts and the meaning of every v3 entry point are unchanged; gce_sam_to_bam / gce_bam_to_sam and SAM text in gce_run_bam were ADDED under v3) */
#define GCE_NONE 0xFFFFFFFFu           /* "no record" marker in uint32 index arrays */
#define GCE_MAX_SUPPORTING_READS 100   /* src/stats.h:15 MAX_SUPPORTING_READS */

typedef struct gce_engine gce_engine;   /* opaque */

typedef enum gce_status {
    GCE_OK = 0,
    GCE_ERR_INVALID = -1,          /* bad argument / bad call order */
    GCE_ERR_NO_DEVICE = -2,        /* no HIP device: the product path has NO CPU fallback */
    GCE_ERR_HIP = -3,              /* HIP runtime failure, see gce_last_error() */
    GCE_ERR_OOM = -4,
    /* fatal conditions of the reference path, reported instead of exit(-1).  When a stream has several, the one on the
     * EARLIEST read is reported (the reference stops at the first in stream order). */
    GCE_ERR_UNSORTED = -10,        /* src/gencore.cpp:233-241 "the input is unsorted" */
    GCE_ERR_UMI_MISMATCH = -11,    /* src/pair.cpp:201-212 "The UMI of a read pair should be identical" */
    GCE_ERR_NM_MISSING = -12,      /* src/group.cpp:532-535: NM dereferenced although absent (segfault in the reference) */
    GCE_ERR_UMI_PARSE = -13,       /* src/bamutil.cpp:47-62: substr(start) with start > length throws in the reference */
    GCE_ERR_QNAME_SHORT = -14,     /* src/bamutil.cpp:343-346 "copyQName ERROR: desitination qname is shorter" */
    GCE_ERR_REF_WINDOW = -15       /* a clustered read lies outside the reference window staged for its contig (gce_set_reference_window) */
} gce_status;

/* One read's fixed-size fields.  Byte-for-byte the BAM alignment core block (SAMv1 section 4.2), i.e. what
 * htslib's bam1_core_t is decoded from; 32 bytes, the "packed key record" of the clustering scan. */
typedef struct gce_core {
    int32_t  tid;       /* bam1_core_t.tid   (refID)                                    */
    int32_t  pos;       /* bam1_core_t.pos   0-based leftmost                           */
    uint8_t  l_qname;   /* strlen(qname)+1 as stored in BAM; the engine applies htslib's in-memory
                           padding to a multiple of 4 (l_extranul) wherever the reference compares
                           name lengths (src/group.cpp:94,117; src/bamutil.cpp:341-346)            */
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
    int32_t flush_period;                   /* the literal 10000 of src/gencore.cpp:321 */
    double  score_percent_req;              /* -a  scorePercentReq                  = 0.8 */
    char    umi_prefix[32];                 /* -u, already resolved ("auto" -> gce_detect_umi_prefix) */
    /* contig lengths of the BAM header: needed by the cross-contig key, src/gencore.cpp:311 */
    int32_t n_targets;
    const uint32_t *target_len;             /* [n_targets], copied by gce_create */
    /* Stream context for contig-granular shards (multi-GPU): this engine sees a contiguous slice of the globally
     * sorted stream.  tick_offset = number of clustered reads before the slice (the reference's static `tick`,
     * src/gencore.cpp:319); trailing_flush != 0 if a flush event occurs after the slice, so clusters still pending
     * at the end of the slice get -d instead of the end-of-file threshold (quirk Q1).  Shards cut INSIDE a contig by
     * cluster key use gce_batch.tick + gce_set_flush_events instead. */
    int64_t tick_offset;
    int32_t trailing_flush;
    /* --quit_after_contig (Options::maxContig, src/options.cpp:10, src/gencore.cpp:243-246): > 0: the read loop ends at the first read whose
     * tid >= max_contig -- that read is still counted by the pre-Stats (addRead comes first, :222) and sorted-checked (:233-241), then it and
     * everything behind it are ignored; what is pending is finished as at the end of the file.  0: off.  (The field was `reserved`, always 0,
     * up to ABI 3: additive.) */
    int32_t max_contig;
} gce_params;

/* A batch of reads in INPUT ORDER (coordinate-sorted), struct-of-arrays.  Offsets are start offsets, lengths
 * come from gce_core (l_qname, n_cigar, l_qseq).  seq is BAM 4-bit packed (high nibble = even base,
 * src/bamutil.cpp:133-147,167-189), qual is raw Phred.  seq/qual are MUTATED IN PLACE where the reference
 * mutates its bam1_t records (src/group.cpp:503-525,555-556, src/cluster.cpp:227-232; the quality rewrite of
 * src/pair.cpp:158-159 persists only in emitted records, see gce_result).
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
    const char     *mi;          /* optional: NUL-terminated MI:Z strings (src/bamutil.cpp:23-38); NULL if unused */
    size_t qname_bytes, cigar_words, seq_bytes, qual_bytes, mi_bytes;   /* total sizes of the blobs */
    /* optional (key-range shards, see gce_set_flush_events): [n] the value the reference's `tick` has right after this
     * read was added (src/gencore.cpp:319-320), counted over the WHOLE stream; ignored for reads that never reach the
     * cluster map.  NULL: the engine counts ticks itself from gce_params.tick_offset. */
    const uint64_t *tick;
} gce_batch;

/* Additive QC counters touched on the path (src/stats.h:47-65; call sites src/gencore.cpp:110,146,222 and
 * src/cluster.cpp:102,136,142,158,162,173,177,185).  All int64, all additive => one RCCL all-reduce(sum). */
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

/* The output of the path: one row per EMITTED record, struct-of-arrays, in the order of the reference's output set
 * (bamComp, src/gencore.h:19-47): ascending (tid, pos, mtid, mpos, isize); where the reference breaks ties by heap
 * address (quirk Q3) the engine uses the input index, so the order is deterministic.
 * Row k means: "input read src[k], with seq/qual replaced by the bytes at seq_off[k]/qual_off[k] of the compact blobs
 * below, its qname replaced by the qname of input read qname_src[k] (BamUtil::copyQName, src/bamutil.cpp:338), NM
 * patched to nm_new[k] if >= 0 (src/group.cpp:570), FR/RR aux bytes appended if >= 0 (src/pair.cpp:57-67)".
 * Host pointers (gce_drain) or device pointers (gce_result_device), owned by the engine until the next
 * gce_process / gce_reset / gce_destroy. */
typedef struct gce_result {
    int64_t         n_reads;     /* reads of the processed stream */
    int64_t         n_out;       /* emitted records */
    const uint32_t *src;         /* [n_out] input read index of the record (the consensus template, or a pass-through read) */
    const uint8_t  *kind;        /* [n_out] 1 = emitted by outputPair (src/gencore.cpp:145);
                                            2 = mate-unmapped pass-through (src/gencore.cpp:307-309) */
    const uint32_t *qname_src;   /* [n_out] == src[k] if the name is unchanged */
    const int32_t  *nm_new;      /* [n_out] -1 = NM untouched, else the byte written at src/group.cpp:570 */
    const int16_t  *fr;          /* [n_out] -1 = no tag, else the FR:C byte (src/pair.cpp:57-61, low byte, quirk Q8) */
    const int16_t  *rr;          /* [n_out] -1 = no tag, else the RR:C byte (src/pair.cpp:62-67) */
    const uint32_t *mate;        /* [n_out] ROW of the other record of the same output Pair, or GCE_NONE */
    const uint64_t *seq_off;     /* [n_out] byte offset of the record's packed bases in `seq` (16-byte aligned) */
    const uint64_t *qual_off;    /* [n_out] byte offset of its qualities in `qual` (16-byte aligned) */
    const uint8_t  *seq;         /* compact blobs: emitted records only */
    const uint8_t  *qual;
    size_t          seq_bytes, qual_bytes;
    gce_stats       pre;         /* mPreStats  deltas */
    gce_stats       post;        /* mPostStats deltas: reads/bases/mismatches are Stats::addRead (src/stats.cpp:101-121)
                                    over ALL emitted records (what writeBam accumulates once every record is written,
                                    src/gencore.cpp:110) — see INTEGRATION.md for the report-before-drain quirk */
} gce_result;

/* Kernel timing of the last gce_process(), HIP events on the engine's stream. */
typedef struct gce_timing {
    double total_ms;             /* first kernel start -> last kernel end */
    double describe_ms;          /* per-read descriptors, UMI slices, pre-Stats (consumed by pairing and the vote; not cluster formation) */
    double cluster_ms;           /* clustering scan (class, key, block-level leaders), the roofline kernel  */
    double csr_ms;               /* rest of cluster formation: tick scan, flush events, leader table, cluster list + member lists */
    double pairing_ms;           /* mate pairing + UMI grouping per cluster */
    double score_ms;             /* Pair::computeScore launches of the fallback path (0 when every group takes the fused vote) */
    double consensus_ms;         /* template pick + column vote (fused kernel + fallbacks) */
    double finish_ms;            /* duplex merge, filter, tags, stats */
    double output_ms;            /* output order + compaction of the emitted records */
    int64_t n_clusters, n_groups, n_pairs;
    int64_t n_leaders;           /* (cluster, scan block) runs the clustering scan handed to the bucket table (0 when ticks come with the batch) */
} gce_timing;

/* Fill *p with the reference defaults (src/options.cpp:4-40). */
void gce_params_default(gce_params *p);

/* "auto" UMI prefix detection on the first read's qname, src/gencore.cpp:207-220.  Writes "", "umi" or "UMI". */
void gce_detect_umi_prefix(const char *first_qname, char out_prefix[32]);

/* Replaces: Gencore::Gencore(Options*) + Cluster(Options*) construction, src/gencore.cpp:7-19. */
int gce_create(const gce_params *params, gce_engine **out);
void gce_destroy(gce_engine *e);

/* Replaces: Reference::instance(opt)->getData(tid, ...) (src/reference.cpp:33-70) as the source of the staged
 * reference.  `nibbles` is one contig in FastaReader's own 4-bit code (A=1,T=2,C=3,G=4, other=0, LOW nibble =
 * even position; src/fastareader.cpp:106-128,139-152), n_bases its length.  Contigs never set behave like
 * "contig not found in the reference" (getData returns NULL).  Host or device pointer. */
int gce_set_reference(gce_engine *e, int32_t tid, const uint8_t *nibbles, int64_t n_bases);
/* Same from upper-cased ASCII bases (one contig of FastaReader::mAllContigs, src/fastareader.h): packed on the GPU
 * (FastaReader::to4bits, src/fastareader.cpp:139-152). */
int gce_set_reference_ascii(gce_engine *e, int32_t tid, const char *bases, int64_t n_bases);
/* Per-shard staging: only the bases [win_start, win_start + n_bases) of a contig of `contig_len` bases, e.g. what the
 * reads of one key-range shard can touch -- an engine for 1/8 of hg19 then holds 1/8 of the reference.  win_start must be even.
 * Reference::getData's end-of-contig rule (src/reference.cpp:40,60) keeps using contig_len.  gce_process fails with
 * GCE_ERR_REF_WINDOW if a clustered read whose isize != 0 does not lie inside the window of its contig. */
int gce_set_reference_window(gce_engine *e, int32_t tid, int64_t contig_len, int64_t win_start, const char *bases, int64_t n_bases);
/* Convenience: pack an upper-cased ASCII contig into that code on the host (src/fastareader.cpp:139). */
void gce_pack_reference(const char *bases, int64_t n_bases, uint8_t *nibbles_out);

/* Key-range shards (multi-GPU cuts inside a contig): the flush events of the WHOLE stream before its first
 * unmapped read — event j is the read on which `tick` reaches (j+1)*flush_period (src/gencore.cpp:319-322) and carries
 * that read's (tid, pos), which bound the flush walk (src/gencore.cpp:333-354).  Used together with gce_batch.tick;
 * host arrays, copied.  n_events = 0 with tick given means "no flush ever fires".  Unmapped reads must follow every
 * mapped read of the stream in this mode. */
int gce_set_flush_events(gce_engine *e, int32_t n_events, const int32_t *ev_tid, const int32_t *ev_pos);

/* Replaces: the per-read loop body Gencore::addToCluster(b) (src/gencore.cpp:272,469-476) for a whole batch.
 * Host buffers are copied to HBM; the caller keeps ownership.  May be called repeatedly; batches are
 * concatenated in call order (they must continue the same sorted stream). */
int gce_submit(gce_engine *e, const gce_batch *batch);
/* Same, but every pointer in *batch is a DEVICE pointer that stays valid until the next gce_submit_device / gce_reset /
 * gce_destroy; seq/qual are mutated in place in the caller's HBM buffers (zero-copy path used by bench.py and by a
 * GPU BAM decoder).  ONE batch per gce_process is zero copy.  Further calls before gce_process append to the stream (round 5): the engine then keeps its
 * own copy -- the first batch and every further one are copied device to device when the second call is made; a batch may be freed as soon as the call that
 * appended it returns, and the in-place mutations hit the engine's copy (results come back through the compact blobs of gce_result either way). */
int gce_submit_device(gce_engine *e, const gce_batch *batch);

/* Replaces: every clusterByUMI call of the stream (src/gencore.cpp:355 periodic, :409 end of file), the
 * outputPair bookkeeping (src/gencore.cpp:145-160) and the ordering of the output set (src/gencore.h:19-47).
 * Runs the whole HIP pipeline over everything submitted since the last process.  A second call without a new
 * submit is an error (the stream was mutated in place). */
int gce_process(gce_engine *e);

/* Replaces: draining csPairs into Gencore::outputPair (src/gencore.cpp:356-360,410-414).  Copies the result
 * table (emitted records only) to host memory owned by the engine. */
int gce_drain(gce_engine *e, gce_result *out);
/* Device-side view of the same table (device pointers). */
int gce_result_device(gce_engine *e, gce_result *out);

/* Streamed submission for a stream whose totals are known up front (a BAM file indexed by gce_bam_open): gce_reserve sizes the
 * HBM copy once, then every gce_submit / gce_submit_async copies its batch straight into place on a separate HIP stream -- the
 * copy of batch k overlaps the caller's preparation of batch k+1 (src/gencore.cpp:205-274 interleaves sam_read1 and
 * addToCluster the same way).  With gce_submit_async the caller's buffers must stay untouched until gce_submit_wait(ticket)
 * or gce_process returns.  MI tags (gce_batch.mi / mi_off) and per-read ticks (gce_batch.tick; every batch of a stream or none) travel on this path too
 * (round 5); a batch that exceeds the reservation makes the buffers grow instead of failing. */
int gce_reserve(gce_engine *e, int64_t n_reads, size_t qname_bytes, size_t cigar_words, size_t seq_bytes, size_t qual_bytes);
int gce_submit_async(gce_engine *e, const gce_batch *batch, int32_t *ticket);
int gce_submit_wait(gce_engine *e, int32_t ticket);

/* Replaces: Stats::statDepth + Bed::statDepth over the stream (src/stats.cpp:57-84, src/bed.cpp:66-81), i.e. the per-base part of
 * mPreStats->addRead (src/gencore.cpp:222, mapped reads only: src/stats.cpp:118-120) and of mPostStats->addRead in writeBam
 * (src/gencore.cpp:110), computed on the GPU from the batch that is resident after gce_process.  coverage_step = Options::coverageStep
 * (src/options.cpp:36).  Regions: the BED file's (tid, start, end) in FILE order (gce_bed_load); a region whose tid is outside the header
 * is ignored, as Bed::loadFromFile drops it (src/bed.cpp:151-166).  Results (host arrays owned by the engine until the next call):
 * depth bins per contig, 1 + target_len / step each (src/stats.cpp:41-47), and base counts per region (BedRegion::mCount). */
typedef struct gce_depth {
    int32_t        n_targets;
    const int64_t *bin_off;       /* [n_targets + 1] first bin of every contig */
    const int64_t *pre_depth;     /* [bin_off[n_targets]]  mPreStats->mGenomeDepth */
    const int64_t *post_depth;    /*                       mPostStats->mGenomeDepth */
    int32_t        n_regions;
    const int64_t *pre_bed;       /* [n_regions] in the order the regions were given */
    const int64_t *post_bed;
} gce_depth;
int gce_depth_stats(gce_engine *e, int32_t coverage_step, int32_t n_regions, const int32_t *region_tid, const int32_t *region_start,
                    const int32_t *region_end, gce_depth *out);

/* Everything the final Stats merge of a multi-GPU run adds up, as ONE int64 buffer in DEVICE memory (counters + histogram + per-contig depth
 * bins + BED region counts; the buffers the reference makes at src/gencore.cpp:181-184 and fills at src/stats.cpp:39-46,56-83, src/bed.cpp:64-79):
 *     [pre Stats: GCE_STATS_WORDS][post Stats: GCE_STATS_WORDS][pre depth: n_bins][post depth: n_bins][pre BED counts: n_regions][post BED counts: n_regions]
 * Every word is additive over key-range shards of one stream: N ranks merge it with one all-reduce(sum) over RCCL (bench.py), N engines of one process with one
 * add per engine (gce_run_bam_depth).  Depth bins as in gce_depth (1 + target_len / coverage_step per contig, contigs back to back; bin_off: host array of
 * n_targets + 1 entries owned by the engine); BED counts in the order the regions were given (a region whose contig is not in the header counts 0).
 * Valid until the next gce_process / gce_depth_stats / gce_stats_payload_device of this engine.  (Addition under ABI v3, round 5.)
 * The payload is COMPLETE when the call returns (the engine's stream is waited for): it may be read from any stream, e.g. by an RCCL all-reduce. */
typedef struct gce_payload_layout {
    int32_t stats_words;          /* 2 * GCE_STATS_WORDS */
    int32_t n_targets;
    int64_t n_bins;               /* depth bins of one block (pre or post) */
    int32_t n_regions;
    int64_t total_words;          /* stats_words + 2 * n_bins + 2 * n_regions */
    const int64_t *bin_off;       /* [n_targets + 1] */
} gce_payload_layout;
int gce_stats_payload_device(gce_engine *e, int32_t coverage_step, int32_t n_regions, const int32_t *region_tid, const int32_t *region_start,
                             const int32_t *region_end, const int64_t **payload, gce_payload_layout *layout);

int gce_get_timing(gce_engine *e, gce_timing *out);
/* The two Stats blocks of the last gce_process in DEVICE memory: 2 x GCE_STATS_WORDS int64, pre then post -- for the final Stats merge
 * of a multi-GPU run (one RCCL all-reduce(sum); all fields are additive, src/stats.h:47-65) without a bounce through the host. */
int gce_stats_device(gce_engine *e, const int64_t **pre_then_post);

/* ------------------------------------------------------------------------------------------------------------------------
 * Multi-GPU planning, on a GPU from the 32-byte key records of the WHOLE sorted stream (host or device pointers).
 * Clusters shard by cluster key (tid, left), not by read position: a right mate follows its mate (src/gencore.cpp:301-303).
 * ------------------------------------------------------------------------------------------------------------------------ */
/* tick_out[i] = the reference's `tick` right after read i was added (src/gencore.cpp:319-320; unchanged by reads that never reach the
 * cluster map) -> gce_batch.tick of a shard; the flush events (src/gencore.cpp:321-322) as malloc'ed host arrays (gce_free) ->
 * gce_set_flush_events.  GCE_ERR_INVALID if a clustered read follows the first unmapped read (not shardable by key). */
int gce_stream_context(int32_t device, const gce_core *core, int64_t n_reads, int32_t flush_period, uint64_t *tick_out,
                       int32_t *n_events, int32_t **ev_tid, int32_t **ev_pos);
/* shard_out[i] in [0, world): mode 0 = contiguous key ranges of equal read count (configs[3]: the cuts fall inside contigs),
 * mode 1 = whole clusters dealt longest-processing-time first with weight reads^2 (configs[4]: ultra-deep hotspots).  world <= 64. */
int gce_plan_shards(int32_t device, const gce_core *core, int64_t n_reads, int32_t world, int32_t mode, int32_t *shard_out);
void gce_free(void *p);
/* Drop all submitted reads/results but keep params, reference and allocations (for repeated bench steps). */
int gce_reset(gce_engine *e);

const char *gce_last_error(const gce_engine *e);   /* human-readable detail of the last failure */
const char *gce_status_message(int status);        /* the reference's message for a status code */
int gce_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------------------
 * Files: the callers and data formats on either side of the path.  Host code over zlib
 * (gencore_amd/csrc/bamio.cpp); htslib is not used.
 * ------------------------------------------------------------------------------------------------------------------------ */
typedef struct gce_bam gce_bam;       /* a BAM file held in memory: inflated stream + record index */
typedef struct gce_fasta gce_fasta;

typedef struct gce_bam_info {
    int32_t             n_targets;    /* bam_hdr_t.n_targets / target_len / target_name (src/gencore.cpp:171,311) */
    const uint32_t     *target_len;
    const char *const  *target_name;
    const char         *text;         /* SAM header text, l_text bytes */
    int64_t             l_text;
    int64_t             n_records;
    uint64_t            qname_bytes, cigar_words, seq_bytes, qual_bytes, mi_bytes;   /* blob totals over all records (gce_reserve) */
    double              read_s, inflate_s, index_s;                                  /* wall time of gce_bam_open's stages */
} gce_bam_info;

/* Replaces: sam_open + sam_hdr_read + the sam_read1 loop (src/gencore.cpp:164-205): reads the file, inflates its BGZF blocks on
 * `threads` host threads (0 = all), parses the header and indexes the records. */
int gce_bam_open(const char *path, int threads, gce_bam **out);
void gce_bam_close(gce_bam *f);
const char *gce_bam_error(const gce_bam *f);
int gce_bam_get_info(const gce_bam *f, gce_bam_info *out);
/* Records [first, first+count) as a gce_batch (host pointers into one of two internal buffer sets, slot = 0 / 1, valid until that
 * slot is filled again): the 32-byte core block verbatim, name / CIGAR / bases / qualities blobs, NM (type + value, bam_aux2i)
 * and MI:Z (src/bamutil.cpp:23-38) from the aux area. */
int gce_bam_chunk(gce_bam *f, int64_t first, int64_t count, int slot, gce_batch *out);
/* Replaces: sam_hdr_write + Gencore::writeBam / sam_write1 for every output record (src/gencore.cpp:85-111,187-190): row k of
 * `res` (HOST pointers: gce_drain) becomes input record src[k] with the row's bases / qualities, the name of record qname_src[k]
 * (BamUtil::copyQName, src/bamutil.cpp:338-364), the NM byte (src/group.cpp:570) and FR / RR appended as aux type 'C'
 * (src/pair.cpp:57-67); BGZF blocks of 0xff00 bytes deflated at `level` on `threads` threads, EOF marker block at the end.
 * level 0..9 = zlib's levels (htslib writes at 6); level -1 = the library's own greedy fixed-Huffman encoder: about 3.5 x the speed of
 * level 1, output about a third larger. */
int gce_bam_write(const char *path, const gce_bam *in, const gce_result *res, int threads, int level);

/* The inverse of gce_bam_chunk: a gce_batch (host pointers) as a BAM file, one record per read with its NM and MI tags.  Not a
 * reference call site: it materialises synthetic streams as files for the end-to-end measurement (tools/bam_bench.py) and tests. */
int gce_bam_from_batch(const char *path, const gce_batch *batch, int32_t n_targets, const uint32_t *target_len,
                       const char *const *target_name, const char *text, int threads, int level);

/* Replaces: the SAM TEXT side of htslib under the reference -- sam_open(in, "r") takes BAM or SAM, sam_read1 parses text lines; an output
 * name that ends in "sam" is opened with sam_open(out, "w") and sam_write1 prints text (src/gencore.cpp:164-173,180,187,205,104).
 * gce_run_bam does the same: an input that does not start with the gzip magic is read as SAM text (header lines, then one alignment per
 * line, converted to BAM records on the host threads and pushed to the GPU like an inflated BAM window), an output name that ends in
 * "sam" is written as text.  The two functions below are the conversion alone, on the host (no engine, no GPU).  Conventions of htslib
 * that the path can see are kept: an integer tag is stored in the smallest type that holds it ('C' for NM 0..255: src/group.cpp:569
 * patches NM only as 'C'), bin = reg2bin(pos, pos + reference length), QUAL '*' = 0xFF bytes, an unknown RNAME = -1. */
int gce_sam_to_bam(const char *sam_path, const char *bam_path, int threads, int level, char err[256]);
int gce_bam_to_sam(const char *bam_path, const char *sam_path, int threads, char err[256]);

/* Replaces: Reference::Reference -> FastaReader(file) + readAll (src/reference.cpp:13-24, src/fastareader.cpp:7-41,57-104,157-168),
 * including its quirks (first character of every line unfiltered, lower case folded, ID = header up to the first blank, a later
 * contig of the same name wins).  Contigs come back as ASCII for gce_set_reference_ascii. */
int gce_fasta_load(const char *path, int threads, gce_fasta **out);   /* threads <= 0: all host cores; 1: the literal one-pass walk */
int gce_fasta_get(const gce_fasta *fa, int32_t *n_contigs, const char *const **ids, const char *const **bases, const int64_t **lengths);
void gce_fasta_free(gce_fasta *fa);

/* Replaces: Bed::loadFromFile (src/bed.cpp:111-168): tab-separated chr / start / end / [name], '#' comment lines skipped, contig names
 * resolved against the BAM header (regions of unknown contigs get tid -1), reading stops at the first line of >= 4095 characters
 * (ifstream::getline into a 4096-byte buffer).  Arrays are malloc'ed; free them with gce_bed_free. */
int gce_bed_load(const char *path, int32_t n_targets, const char *const *target_name, int32_t *n_regions, int32_t **tid, int32_t **start,
                 int32_t **end, char ***name);
void gce_bed_free(int32_t n_regions, int32_t *tid, int32_t *start, int32_t *end, char **name);

typedef struct gce_bam_run {
    int64_t n_reads, n_out;
    double  open_s;          /* gce_bam_open: read + inflate + index */
    double  read_s, inflate_s, index_s;
    double  submit_s;        /* struct-of-arrays fill of every chunk + gce_submit_async (copies overlap the next fill) */
    double  process_s;       /* gce_process wall time (waits for the last copy) */
    double  kernel_ms;       /* gce_timing.total_ms */
    double  drain_s, write_s, total_s;
    gce_stats pre, post;
    int64_t peak_rss_kb;     /* VmHWM of the process when the run ended (the streaming path holds windows, not the file) */
    int64_t rss_start_kb, rss_end_kb;   /* VmRSS when gce_run_bam was entered / left: the path's own footprint is the difference to the peak */
} gce_bam_run;
/* Replaces: Gencore::consensus() end to end for a coordinate-sorted BAM (src/gencore.cpp:162-293) as a pipeline with a bounded host
 * footprint: the file is read and inflated window by window (pinned buffers, copied to HBM while the next window is inflated), records
 * are indexed, parsed and -- after gce_process -- re-assembled as BAM records on the GPU (gce_raw_*), the output stream is deflated by all
 * host threads piece by piece.  chunk_reads < 65536 shrinks the windows to 1 MB of compressed bytes (tests).
 * params->n_targets / target_len are taken from the BAM header; params->umi_prefix "auto" is resolved on the first read
 * (src/gencore.cpp:207-220).  fasta_path may be NULL. */
int gce_run_bam(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int threads,
                int64_t chunk_reads, int level, gce_bam_run *out, char err[256]);
/* The whole-file host-codec path of ABI v2 (gce_bam_open: everything inflated and indexed on the host, struct-of-arrays chunks, gce_bam_write);
 * also what GCE_BAM_HOSTCODEC=1 makes gce_run_bam do. */
int gce_run_bam_hostcodec(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int threads,
                          int64_t chunk_reads, int level, gce_bam_run *out, char err[256]);

/* The GPU side of the BAM codec (gencore_amd/csrc/gce_bamdev.hpp), the pieces gce_run_bam is made of.  Replaces sam_read1's record
 * parsing and sam_write1's record assembly (src/gencore.cpp:205-274,83-111 via htslib) for a whole file at once; the host only inflates
 * and deflates BGZF blocks.
 *   gce_raw_begin / gce_raw_push: the INFLATED BAM stream (header included), in order, window by window; asynchronous (gce_submit_wait on
 *     the ticket before the host buffer is reused; pinned buffers -- gce_host_alloc -- make the copies true DMA).
 *   gce_raw_finish: records indexed and parsed in HBM (records_begin = end of the BAM header, n_ref = its contig count); the engine is then
 *     in the state gce_submit_device leaves it in.  Names, bases and qualities are NOT copied: the batch's blobs are the raw stream.
 *   gce_raw_build_output (after gce_process): the emitted records as a stream of BAM records in HBM, table order; gce_raw_read_output_async
 *     copies a piece of it to the host (ticket -> gce_submit_wait). */
int gce_raw_begin(gce_engine *e, size_t capacity_hint);
int gce_raw_push(gce_engine *e, const void *host, size_t bytes, int32_t *ticket);
/* BGZF members as they lie in the file: copied to HBM COMPRESSED and inflated by the GPU (one lane per member, CRC-32 and ISIZE checked) when
 * gce_raw_finish is called -- replaces bgzf_read's inflate under sam_read1 (src/gencore.cpp:205-274 via htslib); member k lies at comp + coff[k],
 * csize[k] bytes (its BSIZE + 1), usize[k] = its ISIZE.  Their bytes follow what was pushed before.  A damaged member makes gce_raw_finish
 * fail with GCE_ERR_INVALID. */
int gce_raw_push_bgzf(gce_engine *e, const void *comp, size_t comp_bytes, int32_t n_members, const uint64_t *coff, const uint32_t *csize, const uint32_t *usize, int32_t *ticket);
/* The same decoder on the caller's buffers (tests, tools): out receives the members' bytes back to back; first_bad = -1 or the first damaged member. */
int gce_bgzf_inflate(int32_t device, const void *comp, size_t comp_bytes, int32_t n_members, const uint64_t *coff, const uint32_t *csize, const uint32_t *usize, void *out, int32_t *first_bad);
int gce_raw_finish(gce_engine *e, uint64_t records_begin, int32_t n_ref, int64_t *n_records);
int gce_raw_build_output(gce_engine *e, uint64_t *body_bytes, int64_t *n_out);
int gce_raw_read_output_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket);
/* The output record stream (gce_raw_build_output / gce_raw_merge_outputs) compressed into BGZF blocks BY THE GPU -- replaces bgzf_write's deflate
 * under sam_write1 (src/gencore.cpp:104 via htslib): greedy LZ77 + the fixed Huffman codes of RFC 1951, one lane per block, CRC-32 and ISIZE
 * per block (gencore_amd/csrc/gce_deflate.hpp).  *comp_bytes = size of the records' file image; gce_raw_read_deflated_async copies a piece of
 * it out (ticket -> gce_submit_wait).  The caller writes the BAM header's blocks in front and the 28-byte EOF marker behind: what gce_run_bam
 * does for `level == -2` (level -1 = the same scheme on the host's threads; 0..9 = zlib).  gce_bgzf_deflate: the same encoder on the caller's
 * buffer (tests, tools; mirror of gce_bgzf_inflate).  (Additions under ABI v3.) */
int gce_raw_deflate_output(gce_engine *e, uint64_t *comp_bytes);
int gce_raw_read_deflated_async(gce_engine *e, uint64_t offset, void *host, size_t bytes, int32_t *ticket);
int gce_bgzf_deflate(int32_t device, const void *in, size_t n, uint32_t block_bytes, void *out, size_t out_cap, size_t *out_bytes);
int gce_host_alloc(size_t bytes, void **out);
void gce_host_free(void *p);

/* Replaces: Gencore::consensus() (src/gencore.cpp:162-293) for one file over SEVERAL engines, one per entry of `devices` (HIP ordinals; they
 * may repeat), ON THE GPU CODEC: the host reads the file once (BAM or SAM text, as gce_run_bam); every engine receives the compressed pieces
 * (its own PCIe link), inflates and indexes the stream on its own GPU, plans it there (gce_stream_context + gce_plan_shards, plan_mode as
 * there: the same plan on every device, nothing exchanged) and keeps its range of the cluster key with the reads' global ticks and the
 * stream's flush events (gce_raw_select_shard); the engines run side by side; their record streams -- each in bamComp order -- are merged
 * device to device on devices[0] (gce_raw_merge_outputs: order from 32 bytes per record on the host, bytes moved by the GPU), the Stats blocks
 * summed in device memory; one writer.  Same records, same order, same Stats as gce_run_bam.  Needs every mapped read in front of the first
 * unmapped one (a coordinate-sorted file).  params->device / tick_offset / trailing_flush are ignored.  An output name that ends in "sam" is
 * written as text.  GCE_BAM_HOSTCODEC=1 (or gce_run_bam_sharded_hostcodec): round 2's runner -- host inflate / index / cut, per-shard
 * reference windows, host merge; BAM in and out only. */
int gce_run_bam_sharded(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int32_t n_shards,
                        const int32_t *devices, int32_t plan_mode, int threads, int level, gce_bam_run *out, char err[256]);
int gce_run_bam_sharded_hostcodec(const char *in_path, const char *out_path, const char *fasta_path, const gce_params *params, int32_t n_shards,
                                  const int32_t *devices, int32_t plan_mode, int threads, int level, gce_bam_run *out, char err[256]);
/* The pieces of the sharded runner (additions under ABI v3; gencore_amd/csrc/gce_bamdev.hpp).
 *   gce_raw_attach_mirror: from now on every gce_raw_push / gce_raw_push_bgzf to `e` goes to `mirror` as well (same bytes, same tickets);
 *   gce_raw_select_shard (after gce_raw_finish, before gce_process): the engine keeps shard `rank` of `world` of the stream it holds (the
 *     reference's read loop src/gencore.cpp:205-274 restricted to one range of the cluster key, with the tick and the flush walk of the
 *     whole stream, :319-354);
 *   gce_raw_merge_outputs (after every engine's gce_raw_build_output): one record stream in bamComp order (src/gencore.h:19-47 over the
 *     whole file) in engs[0]'s output buffer -- read it with gce_raw_read_output_async(engs[0], ...) --, both Stats blocks summed. */
int gce_raw_attach_mirror(gce_engine *e, gce_engine *mirror);
int gce_raw_select_shard(gce_engine *e, int32_t world, int32_t rank, int32_t plan_mode);
int gce_raw_merge_outputs(gce_engine **engs, int32_t n_engs, uint64_t *body_bytes, int64_t *n_out_total, gce_stats *pre, gce_stats *post, int64_t *n_reads_total);
/*   gce_stats_payload_sum (after every engine's gce_stats_payload_device with the same step and regions): the payloads of engs[1..] added into
 *     engs[0]'s in device memory (device-to-device copies, no host bounce, no collective): the merged depth / BED vectors of one file over several engines. */
int gce_stats_payload_sum(gce_engine **engs, int32_t n_engs, const int64_t **payload, gce_payload_layout *layout);
/* The payload (an engine's own, or the sum on engs[0]) copied to the host: n_words int64. */
int gce_stats_payload_read(gce_engine *e, const int64_t *payload, int64_t n_words, int64_t *host);

/* gce_run_bam / gce_run_bam_sharded WITH the depth statistics of the reference's report (Options::coverageStep, Options::bedFile; src/stats.cpp:56-83,
 * src/bed.cpp:64-79,111-168): n_shards == 1 runs one engine on devices[0] (devices NULL: params->device), n_shards > 1 the sharded runner on the GPU
 * codec.  After the consensus run every engine computes its Stats payload on its GPU (gce_stats_payload_device: the reads it holds are resident),
 * the payloads are summed in device memory (gce_stats_payload_sum) and come back here once.  bed_path may be NULL (no regions).  The arrays of
 * `depth` are malloc'ed: gce_depth_run_free.  depth->pre / post are the Stats blocks AS THEY WERE SUMMED IN THE PAYLOAD (equal to out->pre / out->post). */
typedef struct gce_depth_run {
    int32_t  n_targets;
    int64_t  n_bins;
    int64_t *bin_off;             /* [n_targets + 1] */
    int64_t *pre_depth, *post_depth;      /* [n_bins] */
    int32_t  n_regions;
    int32_t *region_tid, *region_start, *region_end;   /* the BED file's regions in file order (gce_bed_load) */
    int64_t *pre_bed, *post_bed;          /* [n_regions] */
    gce_stats pre, post;
    int64_t  payload_bytes;       /* size of the one buffer a multi-GPU run merges */
} gce_depth_run;
int gce_run_bam_depth(const char *in_path, const char *out_path, const char *fasta_path, const char *bed_path, int32_t coverage_step, const gce_params *params,
                      int32_t n_shards, const int32_t *devices, int32_t plan_mode, int threads, int level, gce_bam_run *out, gce_depth_run *depth, char err[256]);
void gce_depth_run_free(gce_depth_run *depth);

#ifdef __cplusplus
}
#endif
#endif /* GENCORE_AMD_H */
