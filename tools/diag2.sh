cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'W',c['waves'],'passes',c['passes'],'wgs',c['workgroups'], c.get('engine'), c.get('group_lanes'), c.get('streams'), 'pad', c['cells_padded_over_real'], d['kernel_ms'])
"
}
run
run --uniform-len 360
run --cols 24 --group 16
run --cols 12 --group 32
run --cols 8 --group 64
run --cols 24 --group 16 --max-waves 8
run --cols 12 --group 32 --max-waves 8
run --cols 12 --group 32 --max-waves 12
run --cols 8 --group 64 --max-waves 8
run --cols 8 --group 64 --max-waves 12
run --engine 1
run --config 3
