#!/bin/bash
# per-kernel average durations of one short bench run: tools/quick_trace.sh [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/qt; rm -rf gpurun_out/qt/*
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qt -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/qt/log.txt 2>&1
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/qt/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name'].split('(')[0]
    if 'k_' in n: print('%-28s calls=%s avg_us=%.1f' % (n[:28], r['Calls'], float(r['AverageNs'])/1e3))
P
