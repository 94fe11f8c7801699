#!/bin/bash
# k_vote's instruction counters for one or more builds (GPU box): tools/kv_counters.sh "abx/a.so abx/b.so" [kernel-name-prefix]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kvc
K=${2:-k_vote}
for L in $1; do
  n=$(basename $L .so); rm -rf gpurun_out/kvc/$n; mkdir -p gpurun_out/kvc/$n
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    t=$(echo $pass | cut -c1-12 | tr ' ' '_')
    GCE_LIB=$PWD/$L timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/kvc/$n/$t -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/kvc/$n/$t.log 2>&1
  done
  python - $n "$K" <<'P'
import csv,glob,sys,collections
n,K=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(float); disp=0
for fn in glob.glob('gpurun_out/kvc/%s/**/*counter_collection.csv'%n,recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if k==K: acc[r['Counter_Name']]+=float(r['Counter_Value'])
nl=2.0  # launches (1 warm-up + 1 step)
print(n, K, ' '.join('%s=%.4g'%(k,v/nl) for k,v in sorted(acc.items())))
P
done
