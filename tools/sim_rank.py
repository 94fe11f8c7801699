#!/usr/bin/env python
"""What rank R of an N-GPU `bench.py --gpus N` run does, on ONE GPU and without torch.distributed: plan the whole cfg4s stream,
materialise the rank's key range, run one engine step with ticks + flush events.  python tools/sim_rank.py RANK WORLD [pairs_per_gpu] [workload: cfg3 (bench.py's default at every N since round 5) | cfg4s]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from gencore_amd import capi, synth  # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
wl = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
per_gpu = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else synth.CONFIGS[wl]["n_pairs"]
dev = torch.device("cuda", 0)
lib = capi.load_library()
t0 = time.time()
data = synth.generate(wl, n_pairs=per_gpu * world, seed=0, device=dev, shard=(rank, world), scale=(0.125 * world if wl == "cfg4s" else 1.0))
torch.cuda.synchronize()
t_gen = time.time() - t0
ctx = data.stream_context
t = data.t
tl = np.asarray(data.target_len, np.uint32)
prm = capi.default_params(device=0, n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix=data.info["umi_prefix"], cluster_size_req=data.info["supporting_reads"])
eng = C.c_void_p()
assert lib.gce_create(C.byref(prm), C.byref(eng)) == 0
for tid, (nib, ln) in enumerate(data.reference):
    assert lib.gce_set_reference(eng, tid, nib.data_ptr(), ln) == 0
et, ep = ctx["ev_tid"], ctx["ev_pos"]
assert lib.gce_set_flush_events(eng, len(et), et.ctypes.data, ep.ctypes.data) == 0
t["qname"] = bench.padded_clone(t["qname"])
for k in range(2):
    b = bench.device_batch(capi, t, data.n_reads, bench.padded_clone(t["seq"]), bench.padded_clone(t["qual"]), ctx["tick"])
    rc = lib.gce_submit_device(eng, C.byref(b)) or lib.gce_process(eng)
    assert rc == 0, lib.gce_last_error(eng).decode()
r = capi.GceResult()
lib.gce_result_device(eng, C.byref(r))
tm = capi.GceTiming()
lib.gce_get_timing(eng, C.byref(tm))
print(dict(rank=rank, world=world, reads=data.n_reads, pairs_this_rank=data.info["n_pairs"], generate_s=round(t_gen, 1), events=len(et), n_out=int(r.n_out),
           total_ms=round(tm.total_ms, 3), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 1)))
