#!/bin/bash
# cumulative THROUGHPUT cost of k_pairing_sub's phases: builds of the kernel that end at tick k (-DPS_STOP=k), each timed alone on the bench workload.
#   tools/pair_stop.sh build (CPU box)      tools/pair_stop.sh run [bench args] (GPU box)
# ticks: 0 members + name records | 1 name words + hash | 2 name classes + verification | 3 order words + rank | 4 setRight + read->pair | 5 UMI grouping | 6 layout + cluster record
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ab
  for k in 0 1 2 3 4 5 6; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPS_STOP=$k gencore_amd/csrc/engine.hip gencore_amd/csrc/bamio.cpp -o ab/pstop$k.so -lz -lpthread 2>/dev/null & done; wait
else
  shift
  for k in 0 1 2 3 4 5 6; do GCE_LIB=$PWD/ab/pstop$k.so python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep "k_pairing_sub up to" | tail -1; done
fi
