# per-configuration rocprofv3 summaries (kernel stats + PMC) and the one-GPU rehearsal of the sharded path
cd $GRAFT_REPO_ROOT
for c in "2 20" "3 10" "5 4" "4 3"; do
  set -- $c
  bash tools/profile_bench.sh $1 $2 > gpurun_out/prof_c$1.log 2>&1
  echo "== config $1"; tail -8 gpurun_out/prof_c$1.log
done
SWG_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 > gpurun_out/r2_bench_dist1.json 2> gpurun_out/r2_bench_dist1.err; echo "dist rc=$?"; tail -2 gpurun_out/r2_bench_dist1.err; head -c 1500 gpurun_out/r2_bench_dist1.json
