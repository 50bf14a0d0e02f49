#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/batch_blocks.txt
: > $O
for bb in 16 32 64 16 32 64; do
  for c in 2 3; do
    timeout -k 10 200 python bench.py --opt batch_blocks=$bb --config $c --steps 40 --warmup 30 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch_blocks $bb config $c:', d['value'], d['kernel_ms']['fill'])" >> $O || exit 1
  done
done
cat $O
timeout -k 10 500 python bench.py --config 5 --whole --steps 2 --warmup 1 > gpurun_out/r4/bench_config5_whole.json 2> gpurun_out/r4/bench_config5_whole.err || { tail -5 gpurun_out/r4/bench_config5_whole.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4/bench_config5_whole.json')); print('config 5 whole:', d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['cells'], d['config']['n_rescored'], d['verify']['ok'], d['roofline']['launches_per_step'], d['cpu_baseline']['value'])"
