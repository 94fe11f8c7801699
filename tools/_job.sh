cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "hand_derived or fuzz_stream or quirk or exotic or smoke or deep" 2>&1 | tail -3 > gpurun_out/r06_p_quick_tests.txt
bash tools/abn.sh "abx/base.so abx/q2sg.so" --workload cfg5 2>&1 | tee gpurun_out/r06_p_ab_cfg5.txt
bash tools/abn.sh "abx/base.so abx/q2sg.so" 2>&1 | tee gpurun_out/r06_p_ab_cfg3.txt
cat gpurun_out/r06_p_quick_tests.txt
