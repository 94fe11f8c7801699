#!/bin/bash
# SQ / LDS counters per kernel at FULL bench size (cfg3 default line) and for cfg5 (deep kernels): tools/lds_round.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  for wl in cfg3 cfg5; do
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq${i}_$wl -o p -- python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_sq${i}_$wl.log 2>&1
  done
done
ls gpurun_out/${TAG}_sq*_cfg3/ | head
