#!/bin/bash
# after the f16 wipe-test change: suite (with the batch test), peptides, configs 2-5, a short fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/check2.txt
: > $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite4.log 2>&1; tail -3 gpurun_out/r4/suite4.log >> $O
grep -q " passed" gpurun_out/r4/suite4.log && ! grep -q " failed" gpurun_out/r4/suite4.log || { tail -30 gpurun_out/r4/suite4.log; exit 1; }
timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 >> $O 2>&1 || { cat $O; exit 1; }
for c in 2 3 4 5; do
  echo "== config $c" >> $O
  timeout -k 10 300 python bench.py --config $c --steps $([ $c -ge 4 ] && echo 4 || echo 40) --warmup $([ $c -ge 4 ] && echo 1 || echo 30) --no-cpu-baseline --no-host-inclusive 2>>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'], (d['configs'][list(d['configs'])[0]].get('rescore') or {}).get('value'))" >> $O || { cat $O; exit 1; }
done
timeout -k 10 300 python tests/fuzz_gpu.py 240 43 > gpurun_out/r4/fuzz2.log 2>&1; tail -1 gpurun_out/r4/fuzz2.log >> $O
cat $O
