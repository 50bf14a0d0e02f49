#!/bin/bash
# config 5's int32 re-score of a short list (12 531 sequences of 8 192 rows on 6 144 lane groups: 2.04 rounds): other
# workgroup sizes move the rounds away from "a little above an integer"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
: > gpurun_out/r4/q32w.txt
for w in 0 12 11 10 9 8 7 6; do
timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --wide16 0 --opt q32_waves=$w --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('q32_waves=$w:', d['value'], d['kernel_ms'], d['config']['n_rescored'])" >> gpurun_out/r4/q32w.txt || exit 1
done
cat gpurun_out/r4/q32w.txt
