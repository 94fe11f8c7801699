#!/bin/bash
# fast GPU loop for kernel work: the quick half of the parity suite, then a short bench with per-kernel times.  tools/gpu_quick.sh [bench args]
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not at_scale and not bam_end_to_end and not cfg4s and not depth_stats and not c_caller and not cfg3-60000" 2>&1 | tail -4
bash tools/quick_trace.sh "$@" 2>&1 | grep -v "k_flag\|k_u64\|k_u32\|k_scan_reduce"
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms/step', d['ms_per_step'], 'scan', r['clustering_scan']['ms'], r['clustering_scan']['frac'], 'formation', r['cluster_formation']['ms'], r['cluster_formation']['frac'], 'leaders', r.get('leader_runs'))
print(r['phase_ms'])"
