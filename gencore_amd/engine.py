"""Host-side mirror of the engine: a thin object wrapper over the C-ABI (include/gencore_amd.h).

Names follow the reference seam: Engine.add_reads ~ Gencore::addToCluster (src/gencore.cpp:469),
Engine.finish ~ Gencore::finishConsensus + the periodic clusterByUMI calls (src/gencore.cpp:355,409),
Engine.output ~ draining csPairs into Gencore::outputPair (src/gencore.cpp:145).
"""
import ctypes as C

import numpy as np

from . import capi
from .batch import ResultTable
from .capi import GceBatch, GceError, GceResult, GceStats, GceTiming


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

    def add_reads(self, batch):
        """Submit a host ReadBatch (copied to HBM)."""
        st = batch.as_struct()
        self._check(self.lib.gce_submit(self._h, C.byref(st)))
        sz = getattr(self, "_sizes", None) or {"seq": 0, "qual": 0}
        self._sizes = {"seq": sz["seq"] + int(batch.seq.size), "qual": sz["qual"] + int(batch.qual.size)}

    def add_reads_device(self, st, keepalive=None):
        """Submit a GceBatch whose pointers are device pointers (zero copy; seq/qual mutated in place)."""
        self._keep = [st, keepalive]
        self._check(self.lib.gce_submit_device(self._h, C.byref(st)))

    def finish(self):
        rc = self.lib.gce_process(self._h)
        self._check(rc)

    def reset(self):
        self._sizes = None
        self._check(self.lib.gce_reset(self._h))

    def timing(self):
        t = GceTiming()
        self._check(self.lib.gce_get_timing(self._h, C.byref(t)))
        return t.as_dict()

    def output(self):
        """Result table as numpy copies (ResultTable)."""
        r = GceResult()
        self._check(self.lib.gce_drain(self._h, C.byref(r)))
        n = r.n_reads

        def arr(ptr, dt, cnt):
            if cnt == 0 or not ptr:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()

        pre, post = GceStats(), GceStats()
        C.memmove(C.byref(pre), C.byref(r.pre), C.sizeof(GceStats))
        C.memmove(C.byref(post), C.byref(r.post), C.sizeof(GceStats))
        seq_bytes = self._result_bytes(r.seq, "seq")
        t = ResultTable(arr(r.out_flag, np.uint8, n), arr(r.qname_src, np.uint32, n), arr(r.nm_new, np.int32, n),
                        arr(r.fr, np.int16, n), arr(r.rr, np.int16, n), arr(r.mate, np.uint32, n),
                        seq_bytes, self._result_bytes(r.qual, "qual"), pre, post)
        t.out_index = arr(r.out_index, np.uint32, r.n_out)
        return t

    def _result_bytes(self, ptr, which):
        n = self._sizes[which]
        if n == 0 or not ptr:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    def run(self, batch, reference=None):
        """Convenience: one whole stream -> ResultTable (the host batch is NOT mutated)."""
        for tid, (nib, ln) in enumerate(reference or []):
            if nib is not None:
                self.set_reference(tid, nib, ln)
        self._sizes = None
        self.add_reads(batch)
        self.finish()
        return self.output()


def run_stream(batch, params, reference=None):
    e = Engine(params)
    try:
        return e.run(batch, reference)
    finally:
        e.close()
