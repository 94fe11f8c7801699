"""Sharding of a sorted stream across GPUs.

Two granularities:
  * shard_by_contig: clusters never span contigs (gencore.cpp:295-312: the key carries the read's own tid), so a contiguous
    range of contigs is an exact shard.  What such a shard must know about the rest of the stream is only the reference's
    global `tick` (gencore.cpp:319): the number of clustered reads before it, and whether a flush fires after it.
  * plan_shards / shard_by_plan: cuts by CLUSTER KEY (tid, left) anywhere inside a contig — a read with isize < 0 belongs to
    the cluster at its mate's position (gencore.cpp:301-303), so a shard's reads interleave with its neighbours' in stream
    order.  Every read then carries its global tick (gce_batch.tick) and every shard gets the flush events of the whole
    stream (gce_set_flush_events): which clusters a flush takes, and hence the UMI threshold a cluster gets (quirk Q1),
    comes out exactly as in the unsharded stream.  mode="range": contiguous key ranges balanced by read count;
    mode="lpt": clusters dealt to the least loaded shard, heaviest first, weighted by depth^2 (ultra-deep hotspots)."""
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


def cluster_left(core):
    """`left` of the cluster key (gencore.cpp:296-303): the mate's position for the right-hand read of a nearby pair."""
    tid, pos, mtid, mpos, isize = (core[k].astype(np.int64) for k in ("tid", "pos", "mtid", "mpos", "isize"))
    near = (mtid == tid) & (np.abs(mpos - pos) < 100000)
    return np.where(near & (isize < 0), mpos, pos)


def stream_context(core, flush_period=10000):
    """(tick, ev_tid, ev_pos) of a whole stream: the value of the reference's `tick` right after each clustered read was
    added (gencore.cpp:319-320) and the reads on which tick % period == 0, i.e. the periodic flushes (gencore.cpp:321-322)
    before the first unmapped read (the one and only finishConsensus comes there, gencore.cpp:255-262)."""
    cm = clustered_mask(core)
    tick = np.cumsum(cm).astype(np.uint64)
    unm = np.nonzero((core["tid"] < 0) | (core["pos"] < 0))[0]
    end = int(unm[0]) if len(unm) else len(core)
    if len(unm) and cm[end:].any():
        raise ValueError("key-range shards need every mapped read before the first unmapped read")
    ev = np.nonzero(cm[:end] & (tick[:end] % np.uint64(flush_period) == 0))[0]
    return tick, core["tid"][ev].astype(np.int32), core["pos"][ev].astype(np.int32)


def plan_shards(core, world, mode="range"):
    """Shard number of every read.  All reads of one cluster key (tid, left) get the same shard; unmapped reads go last."""
    tid = core["tid"].astype(np.int64)
    key = (np.where(tid < 0, np.int64(1) << 30, tid) << 32) | cluster_left(core).clip(0).astype(np.int64)
    if mode == "range":
        srt = np.sort(key)
        cuts = np.asarray([srt[min(len(srt) - 1, (len(srt) * r) // world)] for r in range(1, world)], np.int64) if len(srt) else np.zeros(0, np.int64)
        return np.searchsorted(cuts, key, side="right").astype(np.int32)
    uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    load = np.zeros(world, np.float64)
    owner = np.zeros(len(uk), np.int32)
    for c in np.argsort(-cnt.astype(np.float64) ** 2, kind="stable"):
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += float(cnt[c]) ** 2
    return owner[inv]


def stream_context_gpu(core, flush_period=10000, device=0):
    """stream_context on the GPU through the C-ABI (gce_stream_context): same return values."""
    import ctypes as C
    from . import capi
    lib = capi.load_library()
    core = np.ascontiguousarray(core)
    n = len(core)
    tick = np.zeros(n, np.uint64)
    ne, et, ep = C.c_int32(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    rc = lib.gce_stream_context(device, core.ctypes.data, n, flush_period, tick.ctypes.data, C.byref(ne), C.byref(et), C.byref(ep))
    if rc != 0:
        raise capi.GceError(rc, "gce_stream_context")
    from .engine import _copy_out
    ev_tid, ev_pos = _copy_out(et, np.int32, ne.value), _copy_out(ep, np.int32, ne.value)
    lib.gce_free(et); lib.gce_free(ep)
    return tick, ev_tid, ev_pos


def plan_shards_gpu(core, world, mode="range", device=0):
    """plan_shards on the GPU through the C-ABI (gce_plan_shards)."""
    from . import capi
    lib = capi.load_library()
    core = np.ascontiguousarray(core)
    out = np.zeros(len(core), np.int32)
    rc = lib.gce_plan_shards(device, core.ctypes.data, len(core), world, 0 if mode == "range" else 1, out.ctypes.data)
    if rc != 0:
        raise capi.GceError(rc, "gce_plan_shards")
    return out


def shard_by_plan(batch, plan, rank, tick):
    """Sub-batch of `rank` (reads in stream order) carrying the global ticks; returns (sub_batch, read_indices)."""
    idx = np.nonzero(plan == rank)[0]
    sub = slice_batch(batch, idx)
    sub.tick = np.ascontiguousarray(tick[idx], np.uint64)
    return sub, idx


def cluster_right(core, target_len):
    """`right` of the cluster key (gencore.cpp:304,311); negative or not, the cross-contig form is just a number to the flush walk."""
    tid, pos, mtid, mpos, isize = (core[k].astype(np.int64) for k in ("tid", "pos", "mtid", "mpos", "isize"))
    near = (mtid == tid) & (np.abs(mpos - pos) < 100000)
    tl = np.asarray(target_len, np.int64)
    tlr = np.where((tid >= 0) & (tid < len(tl)), tl[tid.clip(0, max(len(tl) - 1, 0))] if len(tl) else 0, 0)
    return np.where(near, cluster_left(core) + np.abs(isize) - 1, -tlr * (mtid + 1) + mpos)


def contiguous_cuts(core, world, target_len):
    """Read indices at which the sorted stream can be cut into CONTIGUOUS slices that are exact shards with the two scalars
    tick_offset / trailing_flush (like whole contigs) and cost nothing to make (array views): in front of a cut no read may reach
    the first position behind it — neither itself, nor its nearby mate (gencore.cpp:300-304), nor the `right` of its cluster key:
    a flush takes a cluster only once it has passed BOTH ends of the key (gencore.cpp:344-354), so a key with a far right end is
    still open behind the cut.  Returns world + 1 boundaries chosen among the valid cuts next to the equal-read-count quantiles."""
    n = len(core)
    tid, pos, mtid, mpos = (core[k].astype(np.int64) for k in ("tid", "pos", "mtid", "mpos"))
    t = np.where(tid < 0, np.int64(1) << 30, tid)
    near = (mtid == tid) & (np.abs(mpos - pos) < 100000)
    ext = np.maximum(np.where(near, np.maximum(pos, mpos), pos), np.where(clustered_mask(core), cluster_right(core, target_len), pos))
    reach = (t << 32) | ext.clip(0, (1 << 31) - 1)
    key = (t << 32) | pos.clip(0)
    valid = np.ones(n + 1, bool)
    if n:
        valid[1:n] = np.maximum.accumulate(reach)[:-1] < key[1:]
    cand = np.nonzero(valid)[0]
    bounds = [0]
    for r in range(1, world):
        want = (n * r) // world
        k = int(np.searchsorted(cand, want))
        best = cand[min(k, len(cand) - 1)]
        if k > 0 and abs(int(cand[k - 1]) - want) <= abs(int(best) - want):
            best = cand[k - 1]
        bounds.append(max(int(best), bounds[-1]))
    bounds.append(n)
    return bounds


def shard_contiguous(batch, bounds, rank, flush_period=10000):
    """Slice [bounds[rank], bounds[rank+1]) of the stream (views, no copy of the blobs beyond the slice) + its stream context."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    cm = clustered_mask(batch.core)
    before, mine, total = int(cm[:lo].sum()), int(cm[lo:hi].sum()), int(cm.sum())
    ctx = dict(tick_offset=before, trailing_flush=int(total // flush_period > (before + mine) // flush_period))
    return slice_contiguous(batch, lo, hi), np.arange(lo, hi), ctx


def effective_cpus():
    """CPUs this process can actually use: affinity mask, capped by the cgroup-v2 CPU quota (a container may see 256 CPUs and own 16)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return n
