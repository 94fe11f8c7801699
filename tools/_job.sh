cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "hand_derived or fuzz_stream or quirk or exotic or smoke or bam or long" 2>&1 | tail -3 > gpurun_out/r06_u_quick_tests.txt
bash tools/abn.sh "abx/tplane.so abx/og4.so" 2>&1 | tee gpurun_out/r06_u_ab.txt
for L in tplane og4; do for wl in cfg2 cfg5; do GCE_LIB=abx/$L.so python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$L $wl', d['ms_per_step'], d['roofline']['phase_ms']['output_ms'])"; done; done | tee -a gpurun_out/r06_u_ab.txt
cat gpurun_out/r06_u_quick_tests.txt
