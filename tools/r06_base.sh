#!/bin/bash
# round 6 baseline at HEAD (GPU box): bench line, per-kernel times, k_vote phase clocks, SQ / LDS counter passes.   tools/r06_base.sh <tag>
TAG=${1:-r06_a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_cfg3.json
bash tools/quick_trace.sh > gpurun_out/${TAG}_quick_trace.txt 2>&1
GCE_LIB=$PWD/abx/prof.so timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "phases\|k_vote columns" > gpurun_out/${TAG}_vote_prof.txt
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq${i}_cfg3 -o p -- python bench.py --workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_sq${i}_cfg3.log 2>&1
done
cat gpurun_out/${TAG}_bench_cfg3.json gpurun_out/${TAG}_vote_prof.txt; grep "k_vote\|k_pairing\|k_describe\|k_cluster" gpurun_out/${TAG}_quick_trace.txt
