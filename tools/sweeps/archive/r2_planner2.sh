# planner after the occupancy-filter fix: lengths where a 64-lane single pass competes with two 32-lane passes
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
for lq in 1100 1200 1300 1700; do run --lq $lq --nseq 200000 --config 3; run --lq $lq --nseq 200000 --config 3 --autotune; done
run --lq 1200 --nseq 200000 --config 3 --group 32
