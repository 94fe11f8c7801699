cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06_z > gpurun_out/r06_z_profile_round.log 2>&1
bash tools/lds_round.sh r06_z >> gpurun_out/r06_z_profile_round.log 2>&1
tail -3 gpurun_out/r06_z_profile_round.log | cut -c1-300
du -sh gpurun_out
