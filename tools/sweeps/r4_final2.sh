#!/bin/bash
# the round's last validation of the tree (after the list re-run's workgroup change): GPU suite, smoke, the driver's
# bench command + its one-rank rehearsal, kernel stats of the driver's command under rocprofv3, a fuzz soak
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite_final2.log 2>&1; tail -3 gpurun_out/r4/suite_final2.log
grep -q " passed" gpurun_out/r4/suite_final2.log && ! grep -q " failed" gpurun_out/r4/suite_final2.log || { tail -40 gpurun_out/r4/suite_final2.log; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke2.txt 2>&1 || { cat gpurun_out/r4/smoke2.txt; exit 1; }
cat gpurun_out/r4/smoke2.txt
bash tools/sweeps/r4_bench.sh > gpurun_out/r4/final2_bench.txt 2>&1 || { tail -20 gpurun_out/r4/final2_bench.txt; exit 1; }
cat gpurun_out/r4/final2_bench.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_default3 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r4/prof_default3.json 2> $R/gpurun_out/r4/prof_default3.err ) || { tail -5 gpurun_out/r4/prof_default3.err; exit 1; }
echo "default bench under rocprofv3: done"
timeout -k 10 330 python tests/fuzz_gpu.py 280 43 > gpurun_out/r4/fuzz3.log 2>&1; tail -2 gpurun_out/r4/fuzz3.log
grep -q "^OK" gpurun_out/r4/fuzz3.log || exit 1
