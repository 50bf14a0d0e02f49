#!/bin/bash
# the driver's bench command, its one-rank rehearsal and the same command under rocprofv3, against the final traffic table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
bash tools/sweeps/r4_bench.sh > gpurun_out/r4/final3_bench.txt 2>&1 || { tail -20 gpurun_out/r4/final3_bench.txt; exit 1; }
cat gpurun_out/r4/final3_bench.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_default4 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r4/prof_default4.json 2> $R/gpurun_out/r4/prof_default4.err ) || { tail -5 gpurun_out/r4/prof_default4.err; exit 1; }
echo "default bench under rocprofv3: done"
