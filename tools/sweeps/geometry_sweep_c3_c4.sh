cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
run --config 3 --cols 16 --group 32 --max-waves 16
run --config 3 --cols 16 --group 32 --max-waves 12
run --config 3 --cols 16 --group 32 --max-waves 8
run --config 3 --cols 32 --group 16 --max-waves 12
run --config 3 --cols 8 --group 64 --max-waves 16
run --config 3 --cols 24 --group 32 --max-waves 16
export STEPS=3
run --config 4 --cols 24 --group 64 --max-waves 16
run --config 4 --cols 24 --group 64 --max-waves 8
run --config 4 --cols 32 --group 64 --max-waves 12
run --config 4 --cols 12 --group 64 --max-waves 16
run --config 4 --cols 24 --group 32 --max-waves 16
run --config 4 --cols 16 --group 64 --max-waves 16
