cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'resc', c['n_rescored'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
run --config 2
run --config 2 --long-split 1000
run --config 2 --long-split 1500
run --config 2 --long-split 700
run --config 2 --cols 12 --group 32 --max-waves 12
run --config 2 --cols 12 --group 32 --max-waves 16
run --config 2 --cols 24 --group 16 --max-waves 8
run --config 3
STEPS=3 run --config 5
