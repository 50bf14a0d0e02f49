# final evidence of the round: per-configuration rocprofv3 summaries, the default bench line, the
# one-GPU rehearsal of the sharded path, the K=23 wave timeline
cd $GRAFT_REPO_ROOT
for c in "2 20" "3 10" "5 4" "4 3"; do
  set -- $c
  bash tools/profile_bench.sh $1 $2 > gpurun_out/prof_c$1.log 2>&1
  echo "== config $1"; tail -7 gpurun_out/prof_c$1.log | cut -c1-200
done
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"
SWG_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 --steps 5 > gpurun_out/r2_bench_dist_final.json 2> gpurun_out/r2_bench_dist_final.err; echo "dist rc=$?"
rm -f gpurun_out/trace_final2.txt
SWG_TRACE=gpurun_out/trace_final2.txt timeout -k 10 300 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-verify --no-pipeline > /dev/null 2>&1
python tools/trace_timeline.py gpurun_out/trace_final2.txt > gpurun_out/r2_wave_timeline.txt; python tools/trace_hw.py gpurun_out/trace_final2.txt | tail -6 >> gpurun_out/r2_wave_timeline.txt
cat gpurun_out/r2_wave_timeline.txt | cut -c1-220
