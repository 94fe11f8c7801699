"""debug: per-phase block latency of k_vote (GCE_DBG=128 accumulates wall_clock64 deltas of thread 0 into spare Stats slots)"""
import sys, os, ctypes as C
os.environ["GCE_DBG"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from gencore_amd import capi, synth
lib = capi.load_library()
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
d = synth.generate("cfg3", n_pairs=n, device=dev, scale=1.0)
tl = np.asarray(d.target_len, np.uint32)
prm = capi.default_params(n_targets=len(tl), target_len=tl.ctypes.data, umi_prefix="UMI", cluster_size_req=2)
eng = C.c_void_p(); assert lib.gce_create(C.byref(prm), C.byref(eng)) == 0
for tid, (nib, ln) in enumerate(d.reference): lib.gce_set_reference(eng, tid, nib.data_ptr(), ln)
t = d.t; t["qname"] = bench.padded_clone(t["qname"])
b = bench.device_batch(capi, t, d.n_reads, bench.padded_clone(t["seq"]), bench.padded_clone(t["qual"]))
assert lib.gce_submit_device(eng, C.byref(b)) == 0 and lib.gce_process(eng) == 0
r = capi.GceResult(); lib.gce_result_device(eng, C.byref(r))
h = r.post.as_dict()["supporting_hist"]
nb = h[25]
print("groups/blk", h[22]/max(nb,1), "passA items/blk", h[23]/max(nb,1), "contested/blk", h[24]/max(nb,1)); print("blocks", nb, "avg us per phase:", [round(h[26 + j] / max(nb, 1) / 100.0, 2) for j in range(16)])
