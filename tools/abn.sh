#!/bin/bash
# A/B/... several builds of libgencore_amd.so on the same GPU box, interleaved, three rounds: tools/abn.sh "<a.so> <b.so> ..." [bench args...]
LIBS=$1; shift
for rep in 1 2 3; do
  for L in $LIBS; do
    GCE_LIB=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ph=d['roofline']['phase_ms']
print('$L'.split('/')[-1], d['ms_per_step'], ' '.join('%s=%.3f'%(k[:-3],v) for k,v in ph.items()))"
  done
done
