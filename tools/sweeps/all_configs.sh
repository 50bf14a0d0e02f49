cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c.get('engine'), 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),'pad', c['cells_padded_over_real'], 'resc', c['n_rescored'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'], d['roofline']['binding_roof']['frac_of_issue_peak'])
"
}
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run --config 2
run --config 3
STEPS=3 run --config 4
STEPS=3 run --config 5
run --config 1
