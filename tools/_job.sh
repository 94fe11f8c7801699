cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "hand_derived or fuzz_stream or quirk or exotic or smoke" 2>&1 | tail -3 > gpurun_out/r06_t_quick_tests.txt
for rep in 1 2 3; do for v in 1 0; do
  if [ $v = 1 ]; then export GCE_DESCRIBE_SERIAL=1; else unset GCE_DESCRIBE_SERIAL; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ph=d['roofline']['phase_ms']
print('serial=$v', d['ms_per_step'], ' '.join('%s=%.3f'%(k[:-3],v) for k,v in ph.items()))"
done; done 2>&1 | tee gpurun_out/r06_t_ab.txt
unset GCE_DESCRIBE_SERIAL
for wl in cfg2 cfg5; do for v in 1 0; do
  if [ $v = 1 ]; then export GCE_DESCRIBE_SERIAL=1; else unset GCE_DESCRIBE_SERIAL; fi
  python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$wl serial=$v', d['ms_per_step'])"
done; done 2>&1 | tee -a gpurun_out/r06_t_ab.txt
cat gpurun_out/r06_t_quick_tests.txt
