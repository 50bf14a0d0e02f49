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
run
run $B --long-group 64 --long-cols 8
run $B --long-group 32 --long-cols 12
for s in 500 700 1000 1400 2000 3000; do
run $B --long-group 64 --long-cols 8 --long-split $s
run $B --long-group 32 --long-cols 12 --long-split $s
done
