#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/check3.txt
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "batch or work_queue or many_queries or multipass" >> $O 2>&1 || { cat $O; exit 1; }
timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 >> $O 2>&1 || { cat $O; exit 1; }
timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 batch=0 >> $O 2>&1 || { cat $O; exit 1; }
for c in 2 3; do
  timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 30 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c:', d['value'], d['kernel_ms']['fill'])" >> $O || exit 1
done
cat $O
