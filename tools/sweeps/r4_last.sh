#!/bin/bash
# the round's last check of the tree: GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite_last.log 2>&1; tail -3 gpurun_out/r4/suite_last.log
grep -q " passed" gpurun_out/r4/suite_last.log && ! grep -q " failed" gpurun_out/r4/suite_last.log || { tail -40 gpurun_out/r4/suite_last.log; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.txt 2>&1 || { cat gpurun_out/r4/smoke.txt; exit 1; }
cat gpurun_out/r4/smoke.txt
bash tools/sweeps/r4_bench.sh
