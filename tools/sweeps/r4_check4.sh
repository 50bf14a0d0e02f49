#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/check4.txt
: > $O
cp seq-align-gpu_amd/libswg.so /tmp/new.so
for v in before new; do
  if [ $v = new ]; then cp /tmp/new.so seq-align-gpu_amd/libswg.so; else cp seq-align-gpu_amd/libswg_before.so seq-align-gpu_amd/libswg.so; fi
  echo "== $v" >> $O
timeout -k 10 300 python tools/sweeps/r4_multi_q32.py >> $O 2>&1 || { cat $O; exit 1; }
timeout -k 10 200 python bench.py --config 2 --force-bits 32 --steps 20 --warmup 10 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 int32:', d['value'], d['kernel_ms']['fill'])" >> $O || exit 1
timeout -k 10 200 python bench.py --config 2 --gapopen 1 --gapextend -3 --steps 20 --warmup 10 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2 gaps +1/-3:', d['value'], d['kernel_ms']['fill'])" >> $O || exit 1
timeout -k 10 300 python bench.py --config 5 --wide16 0 --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 5 int16 + int32 list:', d['value'], d['kernel_ms'])" >> $O || exit 1
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
cat $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite5.log 2>&1; tail -3 gpurun_out/r4/suite5.log
