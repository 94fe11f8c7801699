#!/bin/bash
# L2-miss traffic (FETCH_SIZE x 2 x 1024 B per the calibration of DESIGN 6, WRITE_SIZE x 1024 B) of the kernels matching a pattern, for several builds under ab/:
#   tools/ktraffic.sh "<lib names under ab/>" [pattern] [bench args]
LIBS=$1; PAT=${2:-k_vote}; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in $LIBS; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/kt_$C; mkdir -p gpurun_out/kt_$C
    GCE_LIB=$PWD/ab/$L.so timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/kt_$C -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/kt_$C/log.txt 2>&1
  done
  python - "$L" "$PAT" <<'P'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for fn in glob.glob('gpurun_out/kt_%s/**/*counter_collection.csv'%C,recursive=True):
        for r in csv.DictReader(open(fn)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if sys.argv[2] in k: acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
tot=0
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',0))[:12]:
    f=v.get('FETCH_SIZE',0)*2*1024/1e9; w=v.get('WRITE_SIZE',0)*1024/1e9; tot+=f+w
    print('%-10s %-24s fetch %.3f GB  write %.3f GB  total %.3f GB' % (sys.argv[1], k[:24], f, w, f+w))
print('%-10s sum over matching kernels: %.3f GB' % (sys.argv[1], tot))
P
done
