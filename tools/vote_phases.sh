#!/bin/bash
# per-phase time, HBM traffic and instruction counts of k_vote (GPU box): tools/vote_phases.sh [pairs]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=${1:-10000000}; KS="0 1 2 3 4 5 6 7 9 10"; O=gpurun_out/vph; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python tools/vote_phases.py $P $KS > $O/log_t.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python tools/vote_phases.py $P $KS > $O/log_f.txt 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python tools/vote_phases.py $P $KS > $O/log_w.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/i -o i -- python tools/vote_phases.py $P $KS > $O/log_i.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c -o c -- python tools/vote_phases.py $P $KS > $O/log_c.txt 2>&1
python tools/vote_phases_summary.py $O $KS | tee $O/summary.csv
tail -3 $O/log_t.txt
