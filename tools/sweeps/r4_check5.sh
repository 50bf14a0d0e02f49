#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/check5.txt
: > $O
for o in "" "engine=2" "engine=1"; do echo "== peptides $o" >> $O; timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 $o 2>&1 | grep "lq\|rror" >> $O || { cat $O; exit 1; }; done
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite6.log 2>&1; tail -3 gpurun_out/r4/suite6.log >> $O
grep -q " passed" gpurun_out/r4/suite6.log && ! grep -q " failed" gpurun_out/r4/suite6.log || { tail -40 gpurun_out/r4/suite6.log; exit 1; }
for c in 2 3; do
  timeout -k 10 200 python bench.py --config $c --steps 40 --warmup 30 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $c:', d['value'], d['kernel_ms']['fill'], d['config']['engine'])" >> $O || exit 1
done
timeout -k 10 300 python tests/fuzz_gpu.py 200 44 > gpurun_out/r4/fuzz3.log 2>&1; tail -1 gpurun_out/r4/fuzz3.log >> $O
cat $O
