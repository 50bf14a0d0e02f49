#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite3.log 2>&1; tail -3 gpurun_out/r4/suite3.log
grep -q " passed" gpurun_out/r4/suite3.log || exit 1
timeout -k 10 300 python tools/sweeps/cli_timing.py > gpurun_out/r4/cli_end_to_end.txt 2>&1 || { tail -20 gpurun_out/r4/cli_end_to_end.txt; exit 1; }
head -48 gpurun_out/r4/cli_end_to_end.txt
# config 4 in 2 wide passes (64 lanes x 32 / 24 columns) against the planner's 6 passes of 16 x 32
for g in "" "--group 64 --cols 32 --max-waves 12" "--group 64 --cols 24 --max-waves 12" "--group 32 --cols 32 --max-waves 12"; do
timeout -k 10 300 python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive $g 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('config 4 [$g]:', d['value'], 'fill', d['kernel_ms']['fill'], 'K',c['cols_per_wave'],'G',c['group_lanes'],'W',c['waves'],'passes',c['passes'],'last',c['last_pass_cols'],'wgs',c['workgroups'])" >> gpurun_out/r4/config4_passes.txt || exit 1
done
cat gpurun_out/r4/config4_passes.txt
