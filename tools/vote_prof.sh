#!/bin/bash
# per-phase block latency of k_vote: builds an instrumented library (-DVB_PROF: thread 0 of every block accumulates wall_clock64
# deltas at the phase boundaries) into ab/prof.so and runs one bench step with it.  Run through gpurun after building here:
#   tools/vote_prof.sh build   (CPU box)      tools/vote_prof.sh run   (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ab
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DVB_PROF gencore_amd/csrc/engine.hip gencore_amd/csrc/bamio.cpp -o ab/prof.so -lz -lpthread
else
  GCE_LIB=$PWD/ab/prof.so python bench.py --workload ${2:-cfg3} --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "phases\|k_deep_prepare\|k_vote columns" | tail -4
fi
