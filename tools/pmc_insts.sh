#!/bin/bash
# dynamic instruction counts per wave for the engine's kernels on a 2M-pair run: tools/pmc_insts.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pi; rm -rf gpurun_out/pi/*
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pi -o p -- python bench.py --pairs 2000000 --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pi/log.txt 2>&1
python - <<'P'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('gpurun_out/pi/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_INSTS_VALU',0))[:8]:
    w=v.get('SQ_WAVES',1) or 1
    print('%-24s waves=%-8d' % (k[:24], w), ' '.join('%s=%.1f' % (c[9:], v[c]/w) for c in sorted(v) if c!='SQ_WAVES'))
P
