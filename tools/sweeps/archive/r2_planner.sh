# planner after the penalty changes: model picks (no autotune) over query lengths, and the bench shapes
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
for lq in 600 700 800 1000 1200 1500 2000; do run --lq $lq --nseq 200000 --config 3; done
run --lq 1000 --nseq 200000 --config 3 --max-waves 4
run --lq 2000 --nseq 200000 --config 3 --autotune
STEPS=60 run --config 2; run --config 3; STEPS=6 run --config 4; run --config 5
