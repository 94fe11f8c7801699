cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --workload cfg5 --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/c5_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_trace -o t -- python bench.py --workload cfg5 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c5_trace.log 2>&1
cut -c1-1500 gpurun_out/c5_bench.json
