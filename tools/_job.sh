cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_o_gpu_suite.log; cat gpurun_out/r06_o_gpu_suite.log
GCE_RAW_TIMING=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "gce_process (device" | head -3
GCE_RAW_TIMING=1 timeout 600 python tools/bam_bench.py --pairs 4000000 --shards 4 --c-caller 2>&1 | grep "gce_process (device\|^{" | tail -8 | cut -c1-400
