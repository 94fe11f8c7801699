#!/bin/bash
# kernel durations of one pytest selection: tools/trace_test.sh "<-k expression>" [name filter]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/bt; rm -rf gpurun_out/bt/*
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bt -o t -- python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$1" > gpurun_out/bt/log.txt 2>&1
grep -E "passed|failed" gpurun_out/bt/log.txt | tail -1
python - "$2" <<'P'
import csv,glob,sys
f=glob.glob("gpurun_out/bt/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r["Name"]
    if sys.argv[1] in n: print(n[:70], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3))
P
