cd $GRAFT_REPO_ROOT
rm -f gpurun_out/trace_r2.txt
SWG_TRACE=gpurun_out/trace_r2.txt timeout -k 10 300 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-verify --no-pipeline > /dev/null 2>&1
python tools/trace_timeline.py gpurun_out/trace_r2.txt
python tools/trace_hw.py gpurun_out/trace_r2.txt | tail -8
bash tools/profile_bench.sh 2 20 > gpurun_out/prof_c2.log 2>&1; tail -8 gpurun_out/prof_c2.log
