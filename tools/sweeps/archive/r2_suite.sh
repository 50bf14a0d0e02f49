# full GPU suite, then the default bench line (all configs + scaling reference)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 && \
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_default.err; head -c 3000 gpurun_out/r2_bench_default.json
