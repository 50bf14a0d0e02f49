cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'resc', c['n_rescored'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'], d['roofline']['binding_roof']['frac_of_measured_issue_peak'])
"
}
run --config 2
run --config 2
run --config 3
STEPS=3 run --config 5
run --config 1
