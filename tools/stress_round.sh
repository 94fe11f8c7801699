#!/bin/bash
# tools/stress_round.sh <seconds> <tag>: tests/stress_shard.py in passes of 4000 iterations (x 8 cases = 32 000 engine lifetimes each) until the time is used up (GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
end=$(( $(date +%s) + ${1:-1200} )); log=gpurun_out/${2:-r04}_stress_shard.log; : > $log; n=0
while [ $(date +%s) -lt $end ]; do
  t0=$(date +%s); timeout 900 python tests/stress_shard.py 4000 2>&1 | grep -v amdgpu.ids | tail -20 >> $log; n=$((n+1))
  echo "pass $n: $(( $(date +%s) - t0 )) s" >> $log
done
grep -c "stress_shard done" $log; grep -E "ITER|second drain" $log | head; tail -3 $log
