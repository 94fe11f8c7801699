#!/bin/bash
# SQ counters per kernel (two passes) on a 2M-pair run; output gpurun_out/<tag>_sq*/
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq$i -o p -- python bench.py --pairs 2000000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_sq$i.log 2>&1
done
