#!/bin/bash
# Collects the judged evidence for a round on the GPU box (run through gpurun):
#   1. bench.py default line (cfg3, 10 M pairs) and the cfg2 line
#   2. rocprofv3 --kernel-trace --stats of the same default command  -> kernel summary
#   3. separate --pmc passes for HBM traffic of the two roofline kernels (FETCH_SIZE, WRITE_SIZE), kernel-trace only
# Output: gpurun_out/<tag>_*   (copy the summaries into profiles/ afterwards)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg3.json
timeout 300 python bench.py --workload cfg2 --cpu-sample-pairs 1000000 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg2.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o t -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_$c.log 2>&1
done
ls gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_FETCH_SIZE 2>/dev/null | head
cat gpurun_out/${TAG}_bench_cfg3.json
