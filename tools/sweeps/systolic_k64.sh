cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c.get('engine'), 'K',c['cols_per_wave'],'W',c['waves'],'P',c['passes'],'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], d['roofline']['binding_roof']['frac_of_measured_issue_peak'])
"
}
run --config 3 --engine 1 --cols 64
run --config 3 --engine 1 --cols 48
run --config 3 --engine 1 --cols 16
run --config 3
STEPS=3 run --config 4 --engine 1 --cols 64
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "geometry_does_not_change or golden_through_search" 2>&1 | tail -2
