#!/bin/bash
# config 5 (8192-aa query, one GPU's share): the planner's 4 passes of 64 lanes x 32 columns against 8 passes of 32 x 32
# and 16 passes of 16 x 32, at equal cells on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
: > gpurun_out/r4/c5_passes.txt
for g in "" "--group 32 --cols 32 --max-waves 12" "--group 16 --cols 32 --max-waves 12" ""; do
timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-verify $g 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('config 5 [$g]:', d['value'], d['kernel_ms'], 'K', c['cols_per_wave'], 'G', c.get('group_lanes'), 'W', c['waves'], 'passes', c.get('passes'), 'form', c.get('cell_form'))" >> gpurun_out/r4/c5_passes.txt || exit 1
done
cat gpurun_out/r4/c5_passes.txt
