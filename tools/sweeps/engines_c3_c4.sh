cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c.get('engine'), 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
run --config 3
run --config 3 --engine 1
run --config 3 --engine 1 --cols 48
run --config 3 --engine 1 --cols 16
run --config 3 --engine 1 --cols 32 --max-waves 8
STEPS=3 run --config 4
STEPS=3 run --config 4 --engine 1
STEPS=3 run --config 4 --engine 1 --cols 48
run --config 2 --engine 1
