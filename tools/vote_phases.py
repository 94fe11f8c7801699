#!/usr/bin/env python
"""Per-phase cost of k_vote: one process, one cfg3 stream, every ab/vstop<k>.so variant (the kernel cut off behind tick k, tools/vote_stop.sh
build) run for two steps in the order given.  Run it under rocprofv3 (--kernel-trace --stats, or one --pmc pass) and feed the per-dispatch
CSVs to tools/vote_phases_summary.py: the k-th PAIR of k_vote dispatches belongs to the k-th variant.
    python tools/vote_phases.py [pairs] k0 k1 ...        (GPU box)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_batch, padded_clone  # noqa: E402
from gencore_amd import capi, synth  # noqa: E402


def main():
    pairs = int(sys.argv[1])
    ks = sys.argv[2:]
    dev = torch.device("cuda", 0)
    data = synth.generate("cfg3", n_pairs=pairs, seed=0, device=dev, align=1, scale=1.0)
    t = data.t
    tl = np.asarray(data.target_len, np.uint32)
    t["qname"] = padded_clone(t["qname"])
    for k in ks:
        lib = capi.load_library(os.path.join(ROOT, "ab", "vstop%s.so" % k), mode=os.RTLD_LOCAL | os.RTLD_NOW)
        prm = capi.default_params(lib, device=0, n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=data.info["umi_prefix"],
                                  cluster_size_req=data.info["supporting_reads"])
        eng = C.c_void_p()
        assert lib.gce_create(C.byref(prm), C.byref(eng)) == 0
        for tid, (nib, ln) in enumerate(data.reference):
            assert lib.gce_set_reference(eng, tid, nib.data_ptr(), ln) == 0
        for _ in range(2):
            seq, qual = padded_clone(t["seq"]), padded_clone(t["qual"])
            b = device_batch(capi, t, data.n_reads, seq, qual, None)
            assert lib.gce_submit_device(eng, C.byref(b)) == 0
            rc = lib.gce_process(eng)
            torch.cuda.synchronize()
            tm = capi.GceTiming()
            lib.gce_get_timing(eng, C.byref(tm))
            print("vstop", k, "rc", rc, "score_ms %.3f" % tm.as_dict()["score_ms"], flush=True)
        lib.gce_destroy(eng)


if __name__ == "__main__":
    main()
