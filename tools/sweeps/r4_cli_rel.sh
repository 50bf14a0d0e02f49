#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/sweeps/cli_timing.py > gpurun_out/r4/cli_end_to_end.txt 2>&1 || { tail -20 gpurun_out/r4/cli_end_to_end.txt; exit 1; }
cat gpurun_out/r4/cli_end_to_end.txt | head -60
for f in "" "--f16 0"; do
timeout -k 10 300 python bench.py --config 7 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive $f 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('relatives $f:', d['value'], d['kernel_ms'], d['config']['cells'], d['config']['n_rescored'])" || exit 1
done
