cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh 3 20 config3_f16 > gpurun_out/prof_c3.log 2>&1
bash tools/profile_bench.sh 2 40 config2_f16 > gpurun_out/prof_c2.log 2>&1
tail -4 gpurun_out/prof_c3.log
