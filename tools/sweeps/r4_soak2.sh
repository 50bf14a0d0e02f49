#!/bin/bash
# soaks of the final tree: single-query fuzz (other seed) and the query-batch fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 460 python tests/fuzz_gpu.py 400 4711 > gpurun_out/r4/fuzz4.log 2>&1; tail -2 gpurun_out/r4/fuzz4.log
grep -q "^OK" gpurun_out/r4/fuzz4.log || exit 1
timeout -k 10 400 python tests/fuzz_multi_gpu.py 300 4712 > gpurun_out/r4/fuzz_multi4.log 2>&1; tail -2 gpurun_out/r4/fuzz_multi4.log
grep -q "^OK" gpurun_out/r4/fuzz_multi4.log || exit 1
