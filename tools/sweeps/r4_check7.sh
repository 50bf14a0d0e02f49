#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/sweeps/r4_multi_q32.py > gpurun_out/r4/multi_after.txt 2>&1 || { cat gpurun_out/r4/multi_after.txt; exit 1; }
cat gpurun_out/r4/multi_after.txt
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite8.log 2>&1; tail -3 gpurun_out/r4/suite8.log
grep -q " passed" gpurun_out/r4/suite8.log && ! grep -q " failed" gpurun_out/r4/suite8.log || { tail -40 gpurun_out/r4/suite8.log; exit 1; }
timeout -k 10 700 python tests/fuzz_gpu.py 600 101 > gpurun_out/r4/fuzz_soak1.log 2>&1; tail -1 gpurun_out/r4/fuzz_soak1.log
