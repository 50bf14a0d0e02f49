cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or multipass or wide16 or high_similarity or full_size_databases" 2>&1 | tail -4 || exit 1
cp seq-align-gpu_amd/libswg.so /tmp/new.so
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'P',c['passes'],'fill', d['kernel_ms']['fill'])
"
}
for v in prev new prev new; do
  if [ $v = new ]; then cp /tmp/new.so seq-align-gpu_amd/libswg.so; else cp seq-align-gpu_amd/libswg_$v.so seq-align-gpu_amd/libswg.so; fi
  echo "== $v"; STEPS=4 run --config 4; STEPS=5 run --config 5; STEPS=20 run --config 3; STEPS=20 run --config 2
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
