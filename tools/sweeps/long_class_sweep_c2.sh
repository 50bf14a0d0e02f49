cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
B="--cols 24 --group 16 --max-waves 4"
run $B --long-cols 8
run $B --long-cols 6
run $B --long-cols 12
run $B --long-cols 8 --long-split 1200
run $B --long-cols 6 --long-split 1200
run $B --long-cols 8 --long-split 2000
run $B --long-cols 6 --long-split 2000
B="--cols 12 --group 32 --max-waves 4"
run $B --long-cols 8
run $B --long-cols 6
run $B --long-cols 6 --long-split 1500
run $B --long-cols 8 --long-split 1500
