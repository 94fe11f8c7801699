#!/bin/bash
# arbitrary PMC groups (one rocprofv3 pass each, --kernel-trace only) for selected kernels on the full cfg3 run
# usage: tools/pmc_groups.sh "<kernel regex>" "<group1 counters>" "<group2 counters>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pg; rm -rf gpurun_out/pg/*
KRE=$1; shift; i=0
for G in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d gpurun_out/pg/g$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pg/g$i.log 2>&1
done
KRE="$KRE" python - <<'P'
import csv,glob,collections,os,re
kre=re.compile(os.environ['KRE'])
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('gpurun_out/pg/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0]
        if kre.search(k): acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    print(k)
    for c in sorted(v): print('   %-36s %.4g' % (c, v[c]))
P
