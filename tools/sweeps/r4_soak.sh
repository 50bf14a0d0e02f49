#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_headline -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-headline > $R/gpurun_out/r4/prof_headline.json 2> $R/gpurun_out/r4/prof_headline.err ) || { tail -5 gpurun_out/r4/prof_headline.err; exit 1; }
echo "headline under rocprofv3: done"
timeout -k 10 460 python tests/fuzz_gpu.py 400 41 > gpurun_out/r4/fuzz1.log 2>&1; tail -2 gpurun_out/r4/fuzz1.log
grep -q "^OK" gpurun_out/r4/fuzz1.log || exit 1
timeout -k 10 400 python tests/fuzz_multi_gpu.py 330 42 > gpurun_out/r4/fuzz_multi1.log 2>&1; tail -2 gpurun_out/r4/fuzz_multi1.log
grep -q "^OK" gpurun_out/r4/fuzz_multi1.log || exit 1
