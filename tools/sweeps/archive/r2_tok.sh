cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 || exit 1
VARIANTS="prev new" STEPS=30 EXTRA="--config 2" bash tools/sweeps/ab.sh
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'P',c['passes'],'fill', d['kernel_ms']['fill'])
"
}
for v in prev new; do
  if [ $v = new ]; then cp /tmp/new.so seq-align-gpu_amd/libswg.so; else cp seq-align-gpu_amd/libswg_$v.so seq-align-gpu_amd/libswg.so; fi
  echo "== $v"; STEPS=4 run --config 4; STEPS=5 run --config 5; STEPS=10 run --config 2 --force-bits 32
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
