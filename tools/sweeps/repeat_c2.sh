cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], d['kernel_ms'])
"
}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "golden_through_search or device_topk or rescore" 2>&1 | tail -2
for i in 1 2 3 4 5 6; do run --config 2; done
run --config 3
