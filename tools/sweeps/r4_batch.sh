#!/bin/bash
# batched queue claims (round 4): peptides, configs 2 and 3, with the batch off / on and other thresholds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/batch_ab.txt
: > $O
for opts in "batch=0" "batch=8" "batch=4" "batch=8 batch_blocks=32" "batch=8 batch_blocks=64"; do
  echo "== peptides $opts" >> $O
  timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 $opts >> $O 2>&1 || exit 1
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "work_queue or golden or short or tiny or many_queries" >> $O 2>&1 || exit 1
for c in 2 3; do
  for b in 0 8; do
    echo "== config $c SWG_OPT_BATCH=$b" >> $O
    timeout -k 10 200 python bench.py --opt batch=$b --config $c --steps 40 --warmup 30 --no-cpu-baseline --no-host-inclusive 2>>$O | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])" >> $O || exit 1
  done
done
cat $O
