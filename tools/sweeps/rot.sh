cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'fill', d['kernel_ms']['fill'])
"
}
cp seq-align-gpu_amd/libswg.so /tmp/new.so
for v in $VARIANTS; do
  if [ $v = new ]; then cp /tmp/new.so seq-align-gpu_amd/libswg.so; else cp seq-align-gpu_amd/libswg_$v.so seq-align-gpu_amd/libswg.so; fi
  echo "== $v"
  run --no-autotune --uniform-len 400 --nseq 100000
  run --no-autotune
  rm -f gpurun_out/trace_hw.txt
  SWG_TRACE=gpurun_out/trace_hw.txt python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-autotune --uniform-len 400 --nseq 100000 > /dev/null 2>&1; python tools/trace_hw.py gpurun_out/trace_hw.txt | grep "rate by rank"
  rm -f gpurun_out/trace_hw.txt
  SWG_TRACE=gpurun_out/trace_hw.txt python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-autotune > /dev/null 2>&1; python tools/trace_hw.py gpurun_out/trace_hw.txt | grep "rate by rank"
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
