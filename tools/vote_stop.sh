#!/bin/bash
# cumulative THROUGHPUT cost of k_vote's phases: builds of the kernel that end at tick k (-DVB_STOP=k), each timed alone on the bench workload.
#   tools/vote_stop.sh build (CPU box)      tools/vote_stop.sh run [bench args] (GPU box)
# ticks: 0 P0 groups | 1 P1+P3 pairs, overlap | 2 P2 classes | 3 voter lists | 4 pass A | 5 pass-B lists | 6 items | 7 decide | 8 | 9 P6 results | 10 write-back
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ab
  for k in 0 1 2 3 4 5 6 7 9 10; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DVB_STOP=$k gencore_amd/csrc/engine.hip gencore_amd/csrc/bamio.cpp -o ab/vstop$k.so -lz -lpthread 2>/dev/null & done; wait
else
  shift
  for k in 0 1 2 3 4 5 6 7 9 10; do GCE_LIB=$PWD/ab/vstop$k.so python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep "k_vote up to" | tail -1; done
fi
