# final tree: GPU suite, smoke, fuzz, default bench line, one-GPU rehearsal of the sharded path
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 && \
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 && \
timeout -k 10 600 python tests/fuzz_gpu.py 240 7 2>&1 | tail -3 || exit 1
fi
SECONDS=0
timeout -k 10 900 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$? in $SECONDS s"
SWG_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 --steps 5 > gpurun_out/r2_bench_dist_final.json 2> gpurun_out/r2_bench_dist_final.err; echo "dist rc=$?"; tail -3 gpurun_out/r2_bench_dist_final.err
