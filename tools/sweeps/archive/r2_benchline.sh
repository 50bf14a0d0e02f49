cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_final2.json 2> gpurun_out/r2_bench_final2.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r2_bench_final2.json)"
SWG_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 --steps 5 > gpurun_out/r2_bench_dist_final2.json 2> gpurun_out/r2_bench_dist_final2.err; echo "dist rc=$? lines=$(wc -l < gpurun_out/r2_bench_dist_final2.json)"
python -c "import __graft_entry__ as g; g.smoke()"
