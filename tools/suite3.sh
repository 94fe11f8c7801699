#!/bin/bash
# the whole GPU suite N times (no -x), every failure with its assertion text: how often does any test fail intermittently?   tools/suite3.sh [N]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( hostname; rocm-smi --showserial 2>/dev/null | grep -i "serial" | head -1 ) > gpurun_out/suite_repeat.log
for i in $(seq 1 ${1:-3}); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR|E  )|passed|failed" | cut -c1-600 | head -40 >> gpurun_out/suite_repeat.log
  echo "--- pass $i done" >> gpurun_out/suite_repeat.log
done
cat gpurun_out/suite_repeat.log
