"""Host-side file formats around the hot path, thin ctypes wrappers over gencore_amd/csrc/bamio.cpp:
BamFile (gce_bam_open / gce_bam_chunk: a sorted BAM -> ReadBatch), write_bam (gce_bam_write), load_fasta (gce_fasta_load) and
run_bam (gce_run_bam: BAM in -> engine -> BAM out, with the wall time of every stage).  No htslib."""
import ctypes as C

import numpy as np

from . import capi
from .batch import ReadBatch
from .capi import CORE_DTYPE, GceBamInfo, GceBamRun, GceBatch, GceError


class BamFile:
    def __init__(self, path, threads=0):
        self.lib = capi.load_library()
        self._h = C.c_void_p()
        rc = self.lib.gce_bam_open(str(path).encode(), threads, C.byref(self._h))
        if rc != 0:
            msg = self.lib.gce_bam_error(self._h).decode() if self._h else ""
            self.close()
            raise GceError(rc, msg)
        self.info = GceBamInfo()
        self.lib.gce_bam_get_info(self._h, C.byref(self.info))
        n = self.info.n_targets
        self.target_len = [int(self.info.target_len[i]) for i in range(n)]
        self.target_name = [self.info.target_name[i].decode() for i in range(n)]
        self.text = C.string_at(self.info.text, self.info.l_text).decode() if self.info.l_text else ""
        self.n_records = int(self.info.n_records)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gce_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def chunk_struct(self, first, count, slot=0):
        b = GceBatch()
        rc = self.lib.gce_bam_chunk(self._h, first, count, slot, C.byref(b))
        if rc != 0:
            raise GceError(rc, "gce_bam_chunk")
        return b

    def batch(self, first=0, count=None):
        """Records [first, first+count) as a ReadBatch (copies out of the reader's buffers)."""
        count = self.n_records - first if count is None else count
        b = self.chunk_struct(first, count)

        def arr(ptr, dt, n):
            if n == 0 or not ptr:
                return np.zeros(0, dt)
            return np.frombuffer(C.string_at(ptr, n * np.dtype(dt).itemsize), dt).copy()
        n = count
        mi = bool(b.mi)
        return ReadBatch(core=arr(b.core, CORE_DTYPE, n), qname_off=arr(b.qname_off, np.uint64, n), qname=arr(b.qname, np.uint8, b.qname_bytes),
                         cigar_off=arr(b.cigar_off, np.uint64, n), cigar=arr(b.cigar, np.uint32, b.cigar_words),
                         seq_off=arr(b.seq_off, np.uint64, n), seq=arr(b.seq, np.uint8, b.seq_bytes),
                         qual_off=arr(b.qual_off, np.uint64, n), qual=arr(b.qual, np.uint8, b.qual_bytes),
                         nm=arr(b.nm, np.int32, n), nm_type=arr(b.nm_type, np.uint8, n),
                         mi_off=arr(b.mi_off, np.uint64, n) if mi else None, mi=arr(b.mi, np.uint8, b.mi_bytes) if mi else None)


def load_fasta(path, threads=0):
    """{contig id: ASCII bases (bytes)} in file order, with FastaReader's quirks (src/fastareader.cpp:57-104).
    threads: 0 = all host cores (the file is cut at provable line starts), 1 = the literal one-pass walk."""
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.gce_fasta_load(str(path).encode(), int(threads), C.byref(h))
    if rc != 0:
        raise GceError(rc, "gce_fasta_load")
    n = C.c_int32()
    ids, seqs, lens = C.POINTER(C.c_char_p)(), C.POINTER(C.c_void_p)(), C.POINTER(C.c_int64)()
    lib.gce_fasta_get(h, C.byref(n), C.byref(ids), C.byref(seqs), C.byref(lens))
    out = {}
    for i in range(n.value):
        out[ids[i].decode()] = C.string_at(seqs[i], lens[i])
    lib.gce_fasta_free(h)
    return out


def run_bam(in_path, out_path, params, fasta=None, threads=0, chunk_reads=1 << 21, level=6):
    """gce_run_bam: returns the GceBamRun record (stage times, Stats blocks)."""
    lib = capi.load_library()
    run = GceBamRun()
    err = (C.c_char * 256)()
    rc = lib.gce_run_bam(str(in_path).encode(), str(out_path).encode(), str(fasta).encode() if fasta else None, C.byref(params), threads,
                         chunk_reads, level, C.byref(run), err)
    if rc != 0:
        raise GceError(rc, err.value.decode(errors="replace"))
    return run


def run_bam_sharded(in_path, out_path, params, devices, fasta=None, plan_mode=0, threads=0, level=6):
    """gce_run_bam_sharded: one engine per entry of `devices` (HIP ordinals, may repeat), planned on the GPU, tables merged."""
    lib = capi.load_library()
    run = GceBamRun()
    err = (C.c_char * 256)()
    dv = (C.c_int32 * len(devices))(*devices)
    rc = lib.gce_run_bam_sharded(str(in_path).encode(), str(out_path).encode(), str(fasta).encode() if fasta else None, C.byref(params), len(devices), dv,
                                 plan_mode, threads, level, C.byref(run), err)
    if rc != 0:
        raise GceError(rc, err.value.decode(errors="replace"))
    return run


def run_bam_depth(in_path, out_path, params, devices, coverage_step, bed=None, fasta=None, plan_mode=0, threads=0, level=6):
    """gce_run_bam_depth: gce_run_bam (one device) / gce_run_bam_sharded (several) with the depth statistics of the reference's report
    (Options::coverageStep, Options::bedFile): returns (run, dict(bin_off, pre_depth, post_depth, regions, pre_bed, post_bed, pre, post, payload_bytes))."""
    from .capi import GceDepthRun
    lib = capi.load_library()
    run, dr = GceBamRun(), GceDepthRun()
    err = (C.c_char * 256)()
    dv = (C.c_int32 * len(devices))(*devices)
    rc = lib.gce_run_bam_depth(str(in_path).encode(), str(out_path).encode(), str(fasta).encode() if fasta else None, str(bed).encode() if bed else None, int(coverage_step),
                               C.byref(params), len(devices), dv, plan_mode, threads, level, C.byref(run), C.byref(dr), err)
    if rc != 0:
        raise GceError(rc, err.value.decode(errors="replace"))
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    nb, nr = int(dr.n_bins), int(dr.n_regions)
    out = dict(bin_off=arr(dr.bin_off, dr.n_targets + 1), pre_depth=arr(dr.pre_depth, nb), post_depth=arr(dr.post_depth, nb),
               regions=[(dr.region_tid[k], dr.region_start[k], dr.region_end[k]) for k in range(nr)], pre_bed=arr(dr.pre_bed, nr), post_bed=arr(dr.post_bed, nr),
               pre=bytes(dr.pre), post=bytes(dr.post), payload_bytes=int(dr.payload_bytes))
    lib.gce_depth_run_free(C.byref(dr))
    return run, out


def write_batch_as_bam(path, batch, target_len, target_name=None, text="@HD\tVN:1.6\tSO:coordinate\n", threads=0, level=1):
    """gce_bam_from_batch: a ReadBatch as a BAM file (synthetic inputs for the end-to-end runs)."""
    lib = capi.load_library()
    st = batch.as_struct()
    tl = np.ascontiguousarray(target_len, np.uint32)
    names = None
    if target_name is not None:
        names = (C.c_char_p * len(tl))(*[n.encode() for n in target_name])
    rc = lib.gce_bam_from_batch(str(path).encode(), C.byref(st), len(tl), tl.ctypes.data, names, text.encode(), threads, level)
    if rc != 0:
        raise GceError(rc, "gce_bam_from_batch")


def sam_to_bam(sam_path, bam_path, threads=0, level=6):
    """gce_sam_to_bam: SAM text -> BAM on the host (what sam_read1 does with text under the reference, src/gencore.cpp:164,205)."""
    lib = capi.load_library()
    err = (C.c_char * 256)()
    rc = lib.gce_sam_to_bam(str(sam_path).encode(), str(bam_path).encode(), threads, level, err)
    if rc != 0:
        raise GceError(rc, err.value.decode(errors="replace"))


def bam_to_sam(bam_path, sam_path, threads=0):
    """gce_bam_to_sam: BAM -> SAM text on the host (what sam_write1 prints for an output name that ends in "sam", src/gencore.cpp:170-173)."""
    lib = capi.load_library()
    err = (C.c_char * 256)()
    rc = lib.gce_bam_to_sam(str(bam_path).encode(), str(sam_path).encode(), threads, err)
    if rc != 0:
        raise GceError(rc, err.value.decode(errors="replace"))


def load_bed(path, target_names):
    """Bed::loadFromFile (src/bed.cpp:111-168): list of (tid, start, end, name) in file order; tid -1 = contig not in the header."""
    lib = capi.load_library()
    names = (C.c_char_p * max(len(target_names), 1))(*[n.encode() for n in target_names])
    n = C.c_int32()
    tid, st, en, nm = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_char_p)()
    rc = lib.gce_bed_load(str(path).encode(), len(target_names), names, C.byref(n), C.byref(tid), C.byref(st), C.byref(en), C.byref(nm))
    if rc != 0:
        raise GceError(rc, "gce_bed_load")
    out = [(tid[k], st[k], en[k], nm[k].decode()) for k in range(n.value)]
    lib.gce_bed_free(n, tid, st, en, nm)
    return out
