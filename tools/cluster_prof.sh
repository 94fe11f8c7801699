#!/bin/bash
# per-phase block latency of k_cluster: builds an instrumented library (-DCL_PROF: thread 0 of every 32nd block accumulates
# wall_clock64 deltas at the phase boundaries) into ab/clprof.so and runs one bench step with it.  Run through gpurun after building here:
#   tools/cluster_prof.sh build   (CPU box)      tools/cluster_prof.sh run   (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCL_PROF gencore_amd/csrc/engine.hip gencore_amd/csrc/bamio.cpp -o ab/clprof.so -lz -lpthread
else
  GCE_LIB=$PWD/ab/clprof.so python bench.py --workload ${2:-cfg3} --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "k_cluster phases" | tail -1
fi
