#!/bin/bash
# tools/stress_lifetimes.sh <lifetimes>: dumps two of tests/stress_shard.py's streams, builds tools/stress_lifetimes.c and runs it on each (GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python - <<'PY'
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np
import fuzzgen
from test_cabi_driver import dump, ascii_of
for seed, kw in ((302, dict(n_mol=60, umi_mode="prefix", period=5)), (11, {})):
    batch, over, reference, contig_len = fuzzgen.make_case(seed, **kw)
    prefix = over.get("umi_prefix", "")
    dump("gpurun_out/life_%d.dump" % seed, batch, contig_len, ascii_of(reference), over.get("cluster_size_req", 1), over.get("flush_period", 10000), prefix)
    print(seed, batch.n, over)
PY
gcc -std=c11 -O1 -Iinclude tools/stress_lifetimes.c -Lgencore_amd/csrc -lgencore_amd -Wl,-rpath,$PWD/gencore_amd/csrc -o gpurun_out/stress_lifetimes || exit 1
for s in 302 11; do timeout 1200 gpurun_out/stress_lifetimes gpurun_out/life_$s.dump ${1:-50000} 2>&1 | grep -v amdgpu.ids | tail -12; done
rm -f gpurun_out/stress_lifetimes gpurun_out/life_*.dump
