for rep in 1 2 3; do for L in head cs; do GCE_LIB=ab/$L.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ph=d['roofline']['phase_ms']
print('$L', d['ms_per_step'], ' '.join('%s=%.3f'%(k[:-3],v) for k,v in ph.items()))"; done; done
for w in cfg2 cfg5; do for L in head cs head cs; do GCE_LIB=ab/$L.so python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w $L', d['ms_per_step'])"; done; done
python -m pytest tests -m gpu -q -x -k "exotic or fuzz_stream or quirk" 2>&1 | tail -2
