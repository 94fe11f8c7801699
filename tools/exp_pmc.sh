cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp2
for x in 1 2 3 4 5 6 0; do
  GCE_EXP=$x timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/exp2/e$x -o p -- python bench.py --pairs 2000000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/exp2/e$x.log 2>&1
  python - <<P
import csv,glob,collections
f=glob.glob('gpurun_out/exp2/e$x/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if r['Kernel_Name'].startswith('k_consensus_lean2'):
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
w=acc.get('SQ_WAVES',1)
print('EXP $x', {k[9:]: round(v/w,1) for k,v in acc.items() if k!='SQ_WAVES'})
P
done
