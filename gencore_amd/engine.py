"""Host-side mirror of the engine: a thin object wrapper over the C-ABI (include/gencore_amd.h).

Names follow the reference seam: Engine.add_reads ~ Gencore::addToCluster (src/gencore.cpp:469),
Engine.finish ~ Gencore::finishConsensus + the periodic clusterByUMI calls (src/gencore.cpp:355,409),
Engine.rows ~ draining csPairs into Gencore::outputPair and the output set (src/gencore.cpp:145, src/gencore.h:19-47).
"""
import ctypes as C

import numpy as np

from . import capi
from .batch import table_from_rows
from .capi import GceError, GceResult, GceStats, GceTiming

_ROW_FIELDS = (("src", np.uint32), ("kind", np.uint8), ("qname_src", np.uint32), ("nm_new", np.int32), ("fr", np.int16),
               ("rr", np.int16), ("mate", np.uint32), ("seq_off", np.uint64), ("qual_off", np.uint64))


def _copy_out(ptr, dt, cnt):
    """cnt elements of dtype dt at a C pointer, as an owned numpy array.  One string_at + frombuffer: np.ctypeslib.as_array on a pointer builds a new
    ctypes array TYPE per call, and with thousands of engine lifetimes per process (the test-suite, the stress scripts) a 64-bit word of the harness's
    own result structs was found decremented once in ~100 000 lifetimes -- nothing of that kind is made here."""
    addr = C.cast(ptr, C.c_void_p).value
    if not addr or cnt <= 0:
        return np.zeros(0, dt)
    return np.frombuffer(C.string_at(addr, int(cnt) * np.dtype(dt).itemsize), dtype=dt).copy()


class Engine:
    def __init__(self, params=None, **overrides):
        self.lib = capi.load_library()
        self.params = params if params is not None else capi.default_params(**overrides)
        self._keep = []
        self._h = C.c_void_p()
        rc = self.lib.gce_create(C.byref(self.params), C.byref(self._h))
        if rc != 0:
            raise GceError(rc, self.lib.gce_status_message(rc).decode())

    def close(self):
        if self._h:
            self.lib.gce_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise GceError(rc, self.lib.gce_last_error(self._h).decode(errors="replace"))

    def set_reference(self, tid, nibbles, n_bases):
        """nibbles: numpy uint8 (host) or an int device pointer, FastaReader 4-bit code."""
        ptr = nibbles if isinstance(nibbles, int) else np.ascontiguousarray(nibbles, np.uint8).ctypes.data
        self._check(self.lib.gce_set_reference(self._h, tid, ptr, n_bases))

    def set_reference_ascii(self, tid, bases):
        """bases: str / bytes / numpy uint8 of upper-cased ASCII; packed to the 4-bit code on the GPU."""
        if isinstance(bases, str):
            bases = bases.encode()
        a = np.frombuffer(bases, np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, np.uint8)
        self._check(self.lib.gce_set_reference_ascii(self._h, tid, a.ctypes.data, int(a.size)))

    def set_reference_window(self, tid, contig_len, win_start, bases):
        """Per-shard staging: only the bases [win_start, win_start + len(bases)) of a contig (gce_set_reference_window)."""
        if isinstance(bases, str):
            bases = bases.encode()
        a = np.frombuffer(bases, np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, np.uint8)
        self._check(self.lib.gce_set_reference_window(self._h, tid, int(contig_len), int(win_start), a.ctypes.data, int(a.size)))

    def set_flush_events(self, ev_tid, ev_pos):
        """Flush events of the whole stream (key-range shards; see gencore_amd/shard.py)."""
        t, p = np.ascontiguousarray(ev_tid, np.int32), np.ascontiguousarray(ev_pos, np.int32)
        self._check(self.lib.gce_set_flush_events(self._h, int(t.size), t.ctypes.data, p.ctypes.data))

    def add_reads(self, batch):
        """Submit a host ReadBatch (copied to HBM)."""
        st = batch.as_struct()
        self._check(self.lib.gce_submit(self._h, C.byref(st)))

    def add_reads_device(self, st, keepalive=None):
        """Submit a GceBatch whose pointers are device pointers (zero copy; seq/qual mutated in place)."""
        self._keep = [st, keepalive]
        self._check(self.lib.gce_submit_device(self._h, C.byref(st)))

    def finish(self):
        self._check(self.lib.gce_process(self._h))

    def reset(self):
        self._check(self.lib.gce_reset(self._h))

    def timing(self):
        t = GceTiming()
        self._check(self.lib.gce_get_timing(self._h, C.byref(t)))
        return t.as_dict()

    def rows(self):
        """The table of emitted records (gce_drain) as numpy copies: (dict of arrays, pre GceStats, post GceStats)."""
        r = GceResult()
        self._check(self.lib.gce_drain(self._h, C.byref(r)))
        n = int(r.n_out)

        def arr(ptr, dt, cnt):
            if cnt == 0 or not ptr:
                return np.zeros(0, dt)
            return _copy_out(ptr, dt, cnt)

        rows = {name: arr(getattr(r, name), dt, n) for name, dt in _ROW_FIELDS}
        rows["seq"] = arr(r.seq, np.uint8, int(r.seq_bytes))
        rows["qual"] = arr(r.qual, np.uint8, int(r.qual_bytes))
        pre, post = GceStats(), GceStats()
        C.memmove(C.byref(pre), C.byref(r.pre), C.sizeof(GceStats))
        C.memmove(C.byref(post), C.byref(r.post), C.sizeof(GceStats))
        return rows, pre, post

    def output(self, batch):
        """Per-read ResultTable over `batch` (the whole submitted stream, host arrays; not mutated)."""
        rows, pre, post = self.rows()
        t = table_from_rows(batch, rows, pre, post)
        t.out_index = np.sort(rows["src"])
        return t

    def depth_stats(self, step, regions):
        """Stats::statDepth / Bed::statDepth of the processed stream on the GPU (gce_depth_stats).  regions: [(tid, start, end), ...] in
        BED file order.  Returns (bin_off, pre_depth, post_depth, pre_bed, post_bed) as numpy int64."""
        from .capi import GceDepth
        reg = np.asarray([r[:3] for r in regions], np.int32).reshape(-1, 3)
        t, a, b = (np.ascontiguousarray(reg[:, k]) for k in range(3))
        d = GceDepth()
        self._check(self.lib.gce_depth_stats(self._h, int(step), len(reg), t.ctypes.data, a.ctypes.data, b.ctypes.data, C.byref(d)))
        nt = d.n_targets
        off = _copy_out(d.bin_off, np.int64, nt + 1)
        nb = int(off[-1])
        get = lambda ptr, n: _copy_out(ptr, np.int64, n) if n else np.zeros(0, np.int64)
        return off, get(d.pre_depth, nb), get(d.post_depth, nb), get(d.pre_bed, len(reg)), get(d.post_bed, len(reg))

    def run(self, batch, reference=None):
        """Convenience: one whole stream -> ResultTable (the host batch is NOT mutated)."""
        for tid, (nib, ln) in enumerate(reference or []):
            if nib is not None:
                self.set_reference(tid, nib, ln)
        self.add_reads(batch)
        self.finish()
        return self.output(batch)


def run_stream(batch, params, reference=None, events=None):
    e = Engine(params)
    try:
        if events is not None:
            e.set_flush_events(*events)
        return e.run(batch, reference)
    finally:
        e.close()
