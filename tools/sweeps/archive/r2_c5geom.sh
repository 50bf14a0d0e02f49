cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'passes',c['passes'],'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], d['verify']['ok'] if 'verify' in d else None)
"
}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tokens" 2>&1 | tail -3
run --config 5
run --config 5 --max-waves 12
run --config 5 --group 32
run --config 5 --group 32 --max-waves 8
run --config 5 --group 16
run --config 5 --cols 24 --group 64 --max-waves 12
run --config 5 --cols 28 --group 64 --max-waves 12
STEPS=6 run --config 4 --group 32
STEPS=6 run --config 4 --group 32 --cols 24
STEPS=6 run --config 4 --max-waves 8
