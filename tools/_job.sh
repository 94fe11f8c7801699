cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "hand_derived or fuzz_stream or quirk or exotic or smoke or umi" 2>&1 | tail -3 > gpurun_out/r06_s_quick_tests.txt
bash tools/abn.sh "abx/tplane.so abx/prec.so" 2>&1 | tee gpurun_out/r06_s_ab.txt
cat gpurun_out/r06_s_quick_tests.txt
