"""Coordinate sharding of a sorted stream across GPUs.

Clusters never span contigs (gencore.cpp:295-312: the key carries the read's own tid), so a contiguous range of
contigs is an exact shard.  What a shard must know about the rest of the stream is only the reference's global
`tick` (gencore.cpp:319): the number of clustered reads before it, and whether a flush fires after it."""
import numpy as np

from .batch import ReadBatch
from .capi import UINT64_MAX


def slice_batch(batch, idx):
    """Sub-batch with the reads `idx` (ascending), blobs re-packed."""
    idx = np.asarray(idx, np.int64)
    core = batch.core[idx].copy()

    def take(off, data, lens, dtype):
        if len(idx) == 0:
            return np.zeros(0, np.uint64), np.zeros(0, dtype)
        new_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        out = np.empty(int(lens.sum()), dtype)
        for k, i in enumerate(idx):
            o, n = int(off[i]), int(lens[k])
            out[int(new_off[k]):int(new_off[k]) + n] = data[o:o + n]
        return new_off, out

    qlen = core["l_qname"].astype(np.int64)
    qoff, qname = take(batch.qname_off, batch.qname, qlen, np.uint8)
    coff, cigar = take(batch.cigar_off, batch.cigar, core["n_cigar"].astype(np.int64), np.uint32)
    soff, seq = take(batch.seq_off, batch.seq, (core["l_qseq"].astype(np.int64) + 1) // 2, np.uint8)
    loff, qual = take(batch.qual_off, batch.qual, core["l_qseq"].astype(np.int64), np.uint8)
    mi_off = mi = None
    if batch.mi is not None:
        parts, offs, pos = [], [], 0
        for i in idx:
            o = int(batch.mi_off[i])
            if o == UINT64_MAX:
                offs.append(UINT64_MAX)
                continue
            e = o
            while batch.mi[e] != 0:
                e += 1
            parts.append(batch.mi[o:e + 1]); offs.append(pos); pos += e + 1 - o
        mi_off = np.asarray(offs, np.uint64)
        mi = np.concatenate(parts).astype(np.uint8) if parts else np.zeros(1, np.uint8)
    return ReadBatch(core=core, qname_off=qoff, qname=qname, cigar_off=coff, cigar=cigar, seq_off=soff, seq=seq, qual_off=loff,
                     qual=qual, nm=batch.nm[idx].copy(), nm_type=batch.nm_type[idx].copy(), mi_off=mi_off, mi=mi)


def slice_contiguous(batch, lo, hi):
    """Sub-batch of the contiguous read range [lo, hi) — vectorised (no per-read python loop)."""
    core = batch.core[lo:hi].copy()
    n = hi - lo

    def cut(off, data, last_len):
        if n == 0:
            return np.zeros(0, np.uint64), data[:0].copy()
        a = int(off[lo]); e = int(off[hi - 1]) + int(last_len)
        return (off[lo:hi] - np.uint64(a)).astype(np.uint64), data[a:e].copy()
    qoff, qname = cut(batch.qname_off, batch.qname, core["l_qname"][-1] if n else 0)
    coff, cigar = cut(batch.cigar_off, batch.cigar, core["n_cigar"][-1] if n else 0)
    soff, seq = cut(batch.seq_off, batch.seq, (int(core["l_qseq"][-1]) + 1) // 2 if n else 0)
    loff, qual = cut(batch.qual_off, batch.qual, core["l_qseq"][-1] if n else 0)
    assert batch.mi is None, "slice_contiguous: MI blobs not supported"
    return ReadBatch(core=core, qname_off=qoff, qname=qname, cigar_off=coff, cigar=cigar, seq_off=soff, seq=seq, qual_off=loff,
                     qual=qual, nm=batch.nm[lo:hi].copy(), nm_type=batch.nm_type[lo:hi].copy(), mi_off=None, mi=None)


def clustered_mask(core):
    """Reads that reach the cluster map and advance `tick` (gencore.cpp:255-271,295-312)."""
    tid, pos, mtid, mpos, flag = (core[k].astype(np.int64) for k in ("tid", "pos", "mtid", "mpos", "flag"))
    mapped = (tid >= 0) & (pos >= 0) & ((flag & 0x900) == 0)
    near = (mtid == tid) & (np.abs(mpos - pos) < 100000)
    return mapped & (near | (mtid >= 0))


def shard_by_contig(batch, world, rank, flush_period=10000):
    """Contiguous contig ranges balanced by read count.  Returns (sub_batch, read_indices, stream-context dict
    for gce_params: tick_offset, trailing_flush).  Unmapped reads (tid < 0, sorted last) go to the last rank."""
    tid = batch.core["tid"].astype(np.int64)
    n_targets = int(tid.max()) + 1 if len(tid) and tid.max() >= 0 else 1
    counts = np.bincount(tid[tid >= 0], minlength=n_targets)
    cum = np.cumsum(counts)
    total = int(cum[-1]) if len(cum) else 0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cum, total * r / world, side="left")) + 1 if total else 0)
    bounds.append(n_targets)
    bounds = np.maximum.accumulate(np.minimum(bounds, n_targets))
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sel = (tid >= lo) & (tid < hi)
    if rank == world - 1:
        sel |= tid < 0
    idx = np.nonzero(sel)[0]
    cm = clustered_mask(batch.core)
    before = int(cm[(tid >= 0) & (tid < lo)].sum())
    mine = int(cm[sel].sum())
    total_ticks = int(cm.sum())
    later_event = (total_ticks // flush_period) > ((before + mine) // flush_period)
    return slice_batch(batch, idx), idx, dict(tick_offset=before, trailing_flush=int(later_event))
