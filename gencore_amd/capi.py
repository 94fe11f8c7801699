"""ctypes mirror of include/gencore_amd.h and the loader of libgencore_amd.so (the HIP engine).

There is NO CPU fallback in this package: if the shared library is missing or no HIP device is present the
calls fail loudly (GceError / OSError).  The CPU oracle lives in /oracle and is test infrastructure only.
"""
import ctypes as C
import os

import numpy as np

GCE_ABI_VERSION = 3
GCE_NONE = 0xFFFFFFFF
GCE_MAX_SUPPORTING_READS = 100
UINT64_MAX = 0xFFFFFFFFFFFFFFFF

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgencore_amd.so")

# numpy view of gce_core == the 32-byte BAM alignment core block
CORE_DTYPE = np.dtype(
    [("tid", "<i4"), ("pos", "<i4"), ("l_qname", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("n_cigar", "<u2"),
     ("flag", "<u2"), ("l_qseq", "<i4"), ("mtid", "<i4"), ("mpos", "<i4"), ("isize", "<i4")])
assert CORE_DTYPE.itemsize == 32


class GceParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("proper_umi_diff_threshold", C.c_int32), ("unproper_umi_diff_threshold", C.c_int32),
        ("duplex_mismatch_threshold", C.c_int32), ("cluster_size_req", C.c_int32),
        ("base_score_req", C.c_int32), ("high_quality", C.c_int32), ("moderate_quality", C.c_int32),
        ("low_quality", C.c_int32), ("score_high", C.c_int32), ("score_moderate", C.c_int32),
        ("score_low", C.c_int32), ("score_bad", C.c_int32),
        ("skip_low_complexity_cluster_threshold", C.c_int32), ("duplex_only", C.c_int32),
        ("disable_duplex", C.c_int32), ("flush_period", C.c_int32),
        ("score_percent_req", C.c_double), ("umi_prefix", C.c_char * 32),
        ("n_targets", C.c_int32), ("target_len", C.c_void_p),
        ("tick_offset", C.c_int64), ("trailing_flush", C.c_int32), ("max_contig", C.c_int32),
    ]


class GceBatch(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("core", C.c_void_p),
        ("qname_off", C.c_void_p), ("qname", C.c_void_p),
        ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
        ("seq_off", C.c_void_p), ("seq", C.c_void_p),
        ("qual_off", C.c_void_p), ("qual", C.c_void_p),
        ("nm", C.c_void_p), ("nm_type", C.c_void_p),
        ("mi_off", C.c_void_p), ("mi", C.c_void_p),
        ("qname_bytes", C.c_size_t), ("cigar_words", C.c_size_t), ("seq_bytes", C.c_size_t),
        ("qual_bytes", C.c_size_t), ("mi_bytes", C.c_size_t),
        ("tick", C.c_void_p),
    ]


class GceStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "reads", "bases", "reads_unmapped", "bases_unmapped", "base_mismatches", "reads_with_mismatches",
        "clusters", "multi_molecule_clusters", "molecules", "molecules_se", "molecules_pe", "sscs", "dcs",
        "uncounted_supporting_reads")] + [("supporting_hist", C.c_int64 * GCE_MAX_SUPPORTING_READS)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["supporting_hist"] = [int(x) for x in self.supporting_hist]
        return d

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.int64).copy()


GCE_STATS_WORDS = 14 + GCE_MAX_SUPPORTING_READS
assert C.sizeof(GceStats) == 8 * GCE_STATS_WORDS


class GceResult(C.Structure):
    """One row per emitted record, in the order of the reference's output set (include/gencore_amd.h)."""
    _fields_ = [
        ("n_reads", C.c_int64), ("n_out", C.c_int64), ("src", C.c_void_p), ("kind", C.c_void_p), ("qname_src", C.c_void_p),
        ("nm_new", C.c_void_p), ("fr", C.c_void_p), ("rr", C.c_void_p), ("mate", C.c_void_p), ("seq_off", C.c_void_p),
        ("qual_off", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p), ("seq_bytes", C.c_size_t), ("qual_bytes", C.c_size_t),
        ("pre", GceStats), ("post", GceStats),
    ]


class GceTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "total_ms", "describe_ms", "cluster_ms", "csr_ms", "pairing_ms", "score_ms", "consensus_ms", "finish_ms", "output_ms")] + [
        ("n_clusters", C.c_int64), ("n_groups", C.c_int64), ("n_pairs", C.c_int64), ("n_leaders", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


STATUS_NAMES = {
    0: "GCE_OK", -1: "GCE_ERR_INVALID", -2: "GCE_ERR_NO_DEVICE", -3: "GCE_ERR_HIP", -4: "GCE_ERR_OOM",
    -10: "GCE_ERR_UNSORTED", -11: "GCE_ERR_UMI_MISMATCH", -12: "GCE_ERR_NM_MISSING", -13: "GCE_ERR_UMI_PARSE",
    -14: "GCE_ERR_QNAME_SHORT", -15: "GCE_ERR_REF_WINDOW"}

EXPORTED_SYMBOLS = [
    "gce_params_default", "gce_detect_umi_prefix", "gce_create", "gce_destroy", "gce_set_reference", "gce_set_reference_ascii", "gce_set_reference_window",
    "gce_pack_reference", "gce_set_flush_events", "gce_submit", "gce_submit_device", "gce_process", "gce_drain", "gce_result_device",
    "gce_get_timing", "gce_reset", "gce_last_error", "gce_status_message", "gce_abi_version",
    "gce_reserve", "gce_submit_async", "gce_submit_wait",
    "gce_bam_open", "gce_bam_close", "gce_bam_error", "gce_bam_get_info", "gce_bam_chunk", "gce_bam_write", "gce_bam_from_batch", "gce_sam_to_bam", "gce_bam_to_sam",
    "gce_fasta_load", "gce_fasta_get", "gce_fasta_free", "gce_run_bam", "gce_run_bam_hostcodec", "gce_run_bam_sharded", "gce_run_bam_sharded_hostcodec", "gce_raw_deflate_output", "gce_raw_read_deflated_async", "gce_bgzf_deflate", "gce_raw_attach_mirror", "gce_raw_select_shard", "gce_raw_merge_outputs", "gce_raw_begin", "gce_raw_push", "gce_raw_push_bgzf", "gce_bgzf_inflate", "gce_raw_finish", "gce_raw_build_output", "gce_raw_read_output_async", "gce_host_alloc", "gce_host_free", "gce_depth_stats", "gce_stats_payload_device", "gce_stats_payload_sum", "gce_stats_payload_read", "gce_run_bam_depth", "gce_depth_run_free", "gce_stats_device", "gce_stream_context", "gce_plan_shards", "gce_free", "gce_bed_load", "gce_bed_free"]


class GceBamInfo(C.Structure):
    _fields_ = [("n_targets", C.c_int32), ("target_len", C.POINTER(C.c_uint32)), ("target_name", C.POINTER(C.c_char_p)),
                ("text", C.c_void_p), ("l_text", C.c_int64), ("n_records", C.c_int64),
                ("qname_bytes", C.c_uint64), ("cigar_words", C.c_uint64), ("seq_bytes", C.c_uint64), ("qual_bytes", C.c_uint64),
                ("mi_bytes", C.c_uint64), ("read_s", C.c_double), ("inflate_s", C.c_double), ("index_s", C.c_double)]


class GceDepth(C.Structure):
    _fields_ = [("n_targets", C.c_int32), ("bin_off", C.POINTER(C.c_int64)), ("pre_depth", C.POINTER(C.c_int64)), ("post_depth", C.POINTER(C.c_int64)),
                ("n_regions", C.c_int32), ("pre_bed", C.POINTER(C.c_int64)), ("post_bed", C.POINTER(C.c_int64))]


class GcePayloadLayout(C.Structure):
    _fields_ = [("stats_words", C.c_int32), ("n_targets", C.c_int32), ("n_bins", C.c_int64), ("n_regions", C.c_int32), ("total_words", C.c_int64),
                ("bin_off", C.POINTER(C.c_int64))]


class GceDepthRun(C.Structure):
    _fields_ = [("n_targets", C.c_int32), ("n_bins", C.c_int64), ("bin_off", C.POINTER(C.c_int64)), ("pre_depth", C.POINTER(C.c_int64)), ("post_depth", C.POINTER(C.c_int64)),
                ("n_regions", C.c_int32), ("region_tid", C.POINTER(C.c_int32)), ("region_start", C.POINTER(C.c_int32)), ("region_end", C.POINTER(C.c_int32)),
                ("pre_bed", C.POINTER(C.c_int64)), ("post_bed", C.POINTER(C.c_int64)), ("pre", GceStats), ("post", GceStats), ("payload_bytes", C.c_int64)]


class GceBamRun(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_out", C.c_int64), ("open_s", C.c_double), ("read_s", C.c_double), ("inflate_s", C.c_double), ("index_s", C.c_double), ("submit_s", C.c_double),
                ("process_s", C.c_double), ("kernel_ms", C.c_double), ("drain_s", C.c_double), ("write_s", C.c_double),
                ("total_s", C.c_double), ("pre", GceStats), ("post", GceStats), ("peak_rss_kb", C.c_int64), ("rss_start_kb", C.c_int64), ("rss_end_kb", C.c_int64)]


class GceError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("%s (%d): %s" % (STATUS_NAMES.get(status, "?"), status, detail))


_lib = None


def load_library(path=None, mode=C.RTLD_GLOBAL):
    """dlopen libgencore_amd.so; raises OSError if it has not been built (no fallback).  mode=os.RTLD_LOCAL keeps several builds of the
    library apart in one process (tools/vote_phases.py): under RTLD_GLOBAL the second build's kernels bind to the first one's."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GCE_LIB") or LIB_PATH          # GCE_LIB: A/B another build of the same ABI (tools/ab.sh)
    if not os.path.exists(p):
        raise OSError("libgencore_amd.so not built at %s — run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    lib = C.CDLL(p, mode=mode)
    lib.gce_params_default.argtypes = [C.POINTER(GceParams)]
    lib.gce_params_default.restype = None
    lib.gce_detect_umi_prefix.argtypes = [C.c_char_p, C.c_char * 32]
    lib.gce_detect_umi_prefix.restype = None
    lib.gce_create.argtypes = [C.POINTER(GceParams), C.POINTER(C.c_void_p)]
    lib.gce_destroy.argtypes = [C.c_void_p]
    lib.gce_destroy.restype = None
    lib.gce_set_reference.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    lib.gce_set_reference_ascii.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    lib.gce_set_reference_window.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
    lib.gce_pack_reference.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
    lib.gce_pack_reference.restype = None
    lib.gce_set_flush_events.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.gce_submit.argtypes = [C.c_void_p, C.POINTER(GceBatch)]
    lib.gce_submit_device.argtypes = [C.c_void_p, C.POINTER(GceBatch)]
    lib.gce_process.argtypes = [C.c_void_p]
    lib.gce_drain.argtypes = [C.c_void_p, C.POINTER(GceResult)]
    lib.gce_result_device.argtypes = [C.c_void_p, C.POINTER(GceResult)]
    lib.gce_get_timing.argtypes = [C.c_void_p, C.POINTER(GceTiming)]
    lib.gce_stats_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.gce_stream_context.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
    lib.gce_plan_shards.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    lib.gce_free.argtypes = [C.c_void_p]
    lib.gce_free.restype = None
    lib.gce_reset.argtypes = [C.c_void_p]
    lib.gce_last_error.argtypes = [C.c_void_p]
    lib.gce_last_error.restype = C.c_char_p
    lib.gce_status_message.argtypes = [C.c_int]
    lib.gce_status_message.restype = C.c_char_p
    lib.gce_abi_version.restype = C.c_int
    lib.gce_reserve.argtypes = [C.c_void_p, C.c_int64, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.gce_submit_async.argtypes = [C.c_void_p, C.POINTER(GceBatch), C.POINTER(C.c_int32)]
    lib.gce_submit_wait.argtypes = [C.c_void_p, C.c_int32]
    lib.gce_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.gce_bam_close.argtypes = [C.c_void_p]
    lib.gce_bam_close.restype = None
    lib.gce_bam_error.argtypes = [C.c_void_p]
    lib.gce_bam_error.restype = C.c_char_p
    lib.gce_bam_get_info.argtypes = [C.c_void_p, C.POINTER(GceBamInfo)]
    lib.gce_bam_chunk.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(GceBatch)]
    lib.gce_bam_write.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(GceResult), C.c_int, C.c_int]
    lib.gce_bam_from_batch.argtypes = [C.c_char_p, C.POINTER(GceBatch), C.c_int32, C.c_void_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_int, C.c_int]
    lib.gce_sam_to_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    lib.gce_bam_to_sam.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    lib.gce_fasta_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.gce_fasta_get.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.POINTER(C.c_int64))]
    lib.gce_fasta_free.argtypes = [C.c_void_p]
    lib.gce_fasta_free.restype = None
    lib.gce_depth_stats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GceDepth)]
    lib.gce_stats_payload_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(GcePayloadLayout)]
    lib.gce_stats_payload_sum.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(GcePayloadLayout)]
    lib.gce_stats_payload_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.gce_run_bam_depth.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(GceParams), C.c_int32, C.c_void_p, C.c_int32, C.c_int, C.c_int,
                                      C.POINTER(GceBamRun), C.POINTER(GceDepthRun), C.c_char_p]
    lib.gce_depth_run_free.argtypes = [C.POINTER(GceDepthRun)]
    lib.gce_depth_run_free.restype = None
    lib.gce_bed_load.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
                                 C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_char_p))]
    lib.gce_bed_free.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gce_bed_free.restype = None
    lib.gce_bgzf_inflate.argtypes = [C.c_int32, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.gce_run_bam_sharded.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(GceParams), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int, C.c_int, C.POINTER(GceBamRun), C.c_char * 256]
    lib.gce_run_bam.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(GceParams), C.c_int, C.c_int64, C.c_int, C.POINTER(GceBamRun), C.c_char * 256]
    if path is None:
        _lib = lib
    return lib


def default_params(lib=None, **overrides):
    """gce_params with the reference defaults (src/options.cpp:4-40); python-side so it works without the .so too."""
    p = GceParams()
    p.abi_version = GCE_ABI_VERSION
    p.device = 0
    p.proper_umi_diff_threshold = 1
    p.unproper_umi_diff_threshold = 0
    p.duplex_mismatch_threshold = 2
    p.cluster_size_req = 1
    p.base_score_req = 6
    p.high_quality, p.moderate_quality, p.low_quality = 30, 20, 15
    p.score_high, p.score_moderate, p.score_low, p.score_bad = 8, 6, 4, 2
    p.skip_low_complexity_cluster_threshold = 1000
    p.duplex_only = 0
    p.disable_duplex = 0
    p.flush_period = 10000
    p.score_percent_req = 0.8
    p.umi_prefix = b""
    p.n_targets = 0
    p.target_len = None
    p.tick_offset = 0
    p.trailing_flush = 0
    for k, v in overrides.items():
        if k == "umi_prefix" and isinstance(v, str):
            v = v.encode()
        setattr(p, k, v)
    return p
