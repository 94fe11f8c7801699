#!/bin/bash
# Collects the judged evidence for a round on the GPU box (run through gpurun):
#   1. bench.py default line (cfg3, 10 M pairs), the cfg2 line and the cfg5 line (deep clusters)
#   2. rocprofv3 --kernel-trace --stats of the default command and of the cfg5 command  -> kernel summaries
#   3. separate --pmc passes for HBM traffic (FETCH_SIZE, WRITE_SIZE), kernel-trace only
#   4. the end-to-end file path (tools/bam_bench.py)
# Output: gpurun_out/<tag>_*   (copy the summaries into profiles/ afterwards: tools/prof_summary.py, hbm_summary.py, sq_summary.py)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg3.json
timeout 300 python bench.py --workload cfg2 --cpu-sample-pairs 1000000 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg2.json
timeout 300 python bench.py --workload cfg5 --cpu-sample-pairs 200000 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg5.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o t -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace5 -o t -- python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/${TAG}_trace5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_$c.log 2>&1
done
for wl in cfg2 cfg5; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_${wl}_pmc_$c -o p -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_${wl}_pmc_$c.log 2>&1
done; done
timeout 600 python tools/bam_bench.py --pairs 4000000 --shards 4 --c-caller 2>&1 | tail -1 > gpurun_out/${TAG}_bam_e2e_cfg3.json
cat gpurun_out/${TAG}_bench_cfg3.json
