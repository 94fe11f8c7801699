#!/bin/bash
# A/B two builds of libgencore_amd.so on the same GPU box, interleaved: tools/ab.sh <a.so> <b.so> [bench args...]
# (build the baseline first: git stash; build; cp libgencore_amd.so /root/repo/gpurun_ab/a.so; git stash pop; build)
A=$1; B=$2; shift 2   # (ab/ is in .gpurunignore: variants that must travel go to abx/, emptied after use)
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    GCE_LIB=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ph=d['roofline']['phase_ms']
print('$L'.split('/')[-1], d['ms_per_step'], ' '.join('%s=%.3f'%(k[:-3],v) for k,v in ph.items()))"
  done
done
