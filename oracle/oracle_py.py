"""ctypes binding of oracle/libgencore_oracle.so — TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py (cpu_baseline leg).  The product package
(gencore_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gencore_amd.batch import ResultTable
from gencore_amd.capi import GceBatch, GceParams, GceStats

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgencore_oracle.so")


class OrcReference(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("data", C.POINTER(C.c_void_p)), ("n_bases", C.POINTER(C.c_int64))]


class OrcResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("out_flag", C.c_void_p), ("qname_src", C.c_void_p), ("nm_new", C.c_void_p),
                ("fr", C.c_void_p), ("rr", C.c_void_p), ("mate", C.c_void_p), ("pre", GceStats), ("post", GceStats),
                ("n_clusters", C.c_int64), ("n_groups", C.c_int64), ("n_pairs", C.c_int64), ("status", C.c_int),
                ("message", C.c_char * 256)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_run.argtypes = [C.POINTER(GceParams), C.POINTER(OrcReference), C.POINTER(GceBatch), C.POINTER(OrcResult)]
        L.orc_run_shard.argtypes = [C.POINTER(GceParams), C.POINTER(OrcReference), C.POINTER(GceBatch), C.c_int32, C.c_void_p, C.c_void_p,
                                    C.POINTER(OrcResult)]
        L.orc_free_result.argtypes = [C.POINTER(OrcResult)]
        L.orc_free_result.restype = None
        L.orc_get_umi.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_umi_diff.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_is_duplex.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_is_part_of.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_ref_offset.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_m_offset_len.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_m_offset_len.restype = None
        L.orc_cigar_rlen.argtypes = [C.c_void_p, C.c_int]
        L.orc_pack_reference.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        L.orc_stat_depth.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
        L.orc_stat_depth.restype = None
        L.orc_bed_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_bed_depth.restype = None
        L.orc_pack_reference.restype = None
        L.orc_reference_base.argtypes = [C.c_void_p, C.c_int64]
        L.orc_reference_base.restype = C.c_char
        _lib = L
    return _lib


def get_umi(name, prefix):
    buf = C.create_string_buffer(512)
    n = lib().orc_get_umi(name.encode(), prefix.encode(), buf, 512)
    return None if n < 0 else buf.value.decode()


def umi_diff(a, b):
    return lib().orc_umi_diff(a.encode(), len(a), b.encode(), len(b))


def is_duplex(a, b):
    return bool(lib().orc_is_duplex(a.encode(), len(a), b.encode(), len(b)))


def is_part_of(part, whole, is_left):
    p, w = np.asarray(part, np.uint32), np.asarray(whole, np.uint32)
    return bool(lib().orc_is_part_of(p.ctypes.data, len(p), w.ctypes.data, len(w), int(is_left)))


def ref_offset(cigar, bampos):
    c = np.asarray(cigar, np.uint32)
    return lib().orc_ref_offset(c.ctypes.data, len(c), bampos)


def pack_reference(bases):
    """ASCII contig -> FastaReader 4-bit code (A=1,T=2,C=3,G=4, low nibble = even position)."""
    b = bases.encode() if isinstance(bases, str) else bytes(bases)
    out = np.zeros((len(b) + 1) // 2, np.uint8)
    lib().orc_pack_reference(b, len(b), out.ctypes.data)
    return out


def make_reference(contigs):
    """contigs: list of (nibble array or None, n_bases).  Returns (OrcReference, keepalive)."""
    n = len(contigs)
    data = (C.c_void_p * max(n, 1))()
    nb = (C.c_int64 * max(n, 1))()
    keep = []
    for i, (arr, ln) in enumerate(contigs):
        if arr is None:
            data[i] = None
            nb[i] = 0
        else:
            a = np.ascontiguousarray(arr, np.uint8)
            keep.append(a)
            data[i] = a.ctypes.data
            nb[i] = ln
    ref = OrcReference(n, C.cast(data, C.POINTER(C.c_void_p)), C.cast(nb, C.POINTER(C.c_int64)))
    return ref, (data, nb, keep)


def run(batch, params, contigs=None, events=None):
    """Run the oracle over a ReadBatch (NOT mutated: works on a copy).  Returns a ResultTable.
    events = (ev_tid, ev_pos) of the whole stream when `batch` is a key-range shard carrying batch.tick."""
    work = batch.copy()
    work.tick = batch.tick
    st = work.as_struct()
    ref, keep = make_reference(contigs or [])
    res = OrcResult()
    if events is not None:
        et, ep = np.ascontiguousarray(events[0], np.int32), np.ascontiguousarray(events[1], np.int32)
        status = lib().orc_run_shard(C.byref(params), C.byref(ref), C.byref(st), int(et.size), et.ctypes.data, ep.ctypes.data, C.byref(res))
    else:
        status = lib().orc_run(C.byref(params), C.byref(ref), C.byref(st), C.byref(res))
    n = work.n

    def arr(ptr, dt):
        if n == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

    pre, post = GceStats(), GceStats()
    C.memmove(C.byref(pre), C.byref(res.pre), C.sizeof(GceStats))
    C.memmove(C.byref(post), C.byref(res.post), C.sizeof(GceStats))
    table = ResultTable(arr(res.out_flag, np.uint8), arr(res.qname_src, np.uint32), arr(res.nm_new, np.int32),
                        arr(res.fr, np.int16), arr(res.rr, np.int16), arr(res.mate, np.uint32), work.seq, work.qual,
                        pre, post, status=status, message=res.message.decode(errors="replace"))
    table.n_clusters, table.n_groups, table.n_pairs = res.n_clusters, res.n_groups, res.n_pairs
    lib().orc_free_result(C.byref(res))
    del keep
    return table


def depth_stats(batch, table, target_len, step, regions):
    """Stats::statDepth over every mapped input read (pre: Stats::addRead, stats.cpp:101-121 called from gencore.cpp:222) and over
    every emitted record (post: writeBam, gencore.cpp:110).  regions: list of (tid, start, end) in BED file order.
    Returns (bin_off, pre_depth, post_depth, pre_bed, post_bed)."""
    L = lib()
    tl = np.asarray(target_len, np.int64)
    nb = 1 + tl // step
    off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    reg = np.asarray(regions, np.int32).reshape(-1, 3)
    by_tid = [np.nonzero(reg[:, 0] == t)[0] for t in range(len(tl))]          # stable: file order inside a contig
    out = []
    for sel in (np.nonzero(batch.core["tid"] >= 0)[0], np.nonzero(table.out_flag)[0]):
        depth = np.zeros(int(off[-1]), np.int64)
        bed = np.zeros(len(reg), np.int64)
        for t in range(len(tl)):
            rs, re_ = np.ascontiguousarray(reg[by_tid[t], 1]), np.ascontiguousarray(reg[by_tid[t], 2])
            rc = np.zeros(len(rs), np.int64)
            dd = depth[off[t]:off[t + 1]]
            for i in sel[batch.core["tid"][sel] == t]:
                c = batch.core[int(i)]
                L.orc_stat_depth(dd.ctypes.data, int(nb[t]), step, int(c["pos"]), int(c["l_qseq"]))
                if len(rs):
                    L.orc_bed_depth(rs.ctypes.data, re_.ctypes.data, rc.ctypes.data, len(rs), int(c["pos"]), int(c["l_qseq"]))
            bed[by_tid[t]] = rc
        out += [depth, bed]
    return off, out[0], out[2], out[1], out[3]
