#!/bin/bash
# an int32 list's passes in ONE launch (q32_one_launch = 1, the default) against a launch per pass (0): parity tests
# first, then config 5's and the stress variant's `rescore` legs both ways on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch or titin or config5_one_gpu_share or q32_geometry or multipass_int32" > gpurun_out/r4/q32one_tests.log 2>&1 || { tail -30 gpurun_out/r4/q32one_tests.log; exit 1; }
tail -3 gpurun_out/r4/q32one_tests.log
: > gpurun_out/r4/q32one.txt
for cfg in 5 6; do
for m in 0 1 0 1; do
timeout -k 10 300 python bench.py --config $cfg --steps 2 --warmup 1 --wide16 0 --opt q32_one_launch=$m --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config $cfg q32_one_launch=$m:', d['value'], d['kernel_ms'], d['config']['n_rescored'])" >> gpurun_out/r4/q32one.txt || exit 1
done
done
cat gpurun_out/r4/q32one.txt
