cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/sim_rank.py 3 8 2>&1 | tail -1 | tee gpurun_out/r06_v_sim_rank_3_of_8_cfg3.log
python tools/sim_rank.py 0 2 2>&1 | tail -1 | tee -a gpurun_out/r06_v_sim_rank_3_of_8_cfg3.log
timeout 600 python -m pytest tests/test_bench_ranks.py tests/test_dist_gloo.py -m gpu -x -q 2>&1 | tail -2
