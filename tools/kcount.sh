#!/bin/bash
# dynamic instruction counts (total, millions, and per wave) of the kernels matching a pattern, for several builds: tools/kcount.sh "<lib names under ab/>" [pattern] [bench args]
LIBS=$1; PAT=${2:-k_vote}; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in $LIBS; do
  rm -rf gpurun_out/kc; mkdir -p gpurun_out/kc
  GCE_LIB=$PWD/ab/$L.so timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/kc -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/kc/log.txt 2>&1
  python - "$L" "$PAT" <<'P'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob('gpurun_out/kc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if sys.argv[2] in k: acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Dispatch_Id'])]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_INSTS_VALU',0))[:10]:
    d=len({x for x in n if x[0]==k}) or 1; w=v.get('SQ_WAVES',1) or 1
    print('%-10s %-22s launches=%d waves=%-8d' % (sys.argv[1], k[:22], d, w/d), ' '.join('%s=%.1fM(%.0f/w)' % (c[9:], v[c]/d/1e6, v[c]/w) for c in sorted(v) if c!='SQ_WAVES'))
P
done
