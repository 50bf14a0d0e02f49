# the whole GPU suite, the fuzz and the default bench line on the library built with the max-ilp scheduler
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 && \
timeout -k 10 600 python tests/fuzz_gpu.py 2>&1 | tail -4 && \
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_ilp.json 2> gpurun_out/r2_bench_ilp.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_ilp.err; head -c 1500 gpurun_out/r2_bench_ilp.json
