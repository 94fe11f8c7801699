cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --nccl-world1 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r06_d_bench_nccl_world1.json; python -c "
import json; d=json.load(open('gpurun_out/r06_d_bench_nccl_world1.json')); print(d['ms_per_step'], d.get('stats_merge_ms_per_step'), d.get('backend'), d['roofline'].get('traffic_source'))"
