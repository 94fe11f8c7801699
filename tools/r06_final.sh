#!/bin/bash
# The round's evidence in one GPU call, SUMMARISED ON THE BOX (the raw rocprofv3 output of a round is > 64 MiB: gpurun does not bring it back): tools/r06_final.sh <tag> <git-head>
#   bench lines (cfg3 with the CPU baseline + parity_checked, cfg2, cfg5), kernel stats, FETCH / WRITE passes -> hbm_traffic_<workload>.json (stamped with csrc_sha16),
#   SQ / LDS passes -> lds_<workload>.json, k_vote phase clocks, the file path.  Everything lands in gpurun_out/prof/: copy it into profiles/.
TAG=${1:-r06_z}; HEAD=${2:-?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/prof; R=/tmp/raw_$TAG; rm -rf $O $R; mkdir -p $O $R
NOTE="sources at $HEAD"
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3.json
timeout 300 python bench.py --workload cfg2 --cpu-sample-pairs 1000000 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg2.json
timeout 300 python bench.py --workload cfg5 --cpu-sample-pairs 200000 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg5.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -o t -- python bench.py --no-cpu-baseline > $R/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace5 -o t -- python bench.py --workload cfg5 --no-cpu-baseline > $R/trace5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace2 -o t -- python bench.py --workload cfg2 --no-cpu-baseline > $R/trace2.log 2>&1
python tools/prof_summary.py $(find $R/trace -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats.csv "python bench.py --no-cpu-baseline (cfg3), $NOTE" > /dev/null
python tools/prof_summary.py $(find $R/trace5 -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_cfg5.csv "python bench.py --workload cfg5 --no-cpu-baseline, $NOTE" > /dev/null
python tools/prof_summary.py $(find $R/trace2 -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_cfg2.csv "python bench.py --workload cfg2 --no-cpu-baseline, $NOTE" > /dev/null
for wl in cfg3 cfg2 cfg5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/${wl}_pmc_$c -o p -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $R/${wl}_pmc_$c.log 2>&1
  done
  python tools/hbm_summary.py $R/$wl $O/${TAG}_$wl "python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline, $NOTE" $wl > /dev/null
done
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  for wl in cfg3 cfg5; do
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/x_sq${i}_$wl -o p -- python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > $R/x_sq${i}_$wl.log 2>&1
  done
done
for wl in cfg3 cfg5; do python tools/lds_summary.py $R/x $O/$TAG $wl "python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline, $NOTE" > /dev/null; done
if [ -f abx/prof.so ]; then GCE_LIB=$PWD/abx/prof.so timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "k_vote phases" | tail -1 > $O/${TAG}_vote_prof.txt; fi
timeout 600 python tools/bam_bench.py --pairs 4000000 --shards 4 --c-caller 2>/dev/null | tail -1 > $O/${TAG}_bam_e2e_cfg3.json
timeout 300 python bench.py --gpus 2 --test-one-gpu --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_gpus2_test_one_gpu.json
timeout 300 python bench.py --nccl-world1 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/${TAG}_bench_nccl_world1.json
rm -rf $R; ls -la $O; head -c 700 $O/${TAG}_bench_cfg3.json
