cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
run --config 2
for t in 1200 1600 2000 2400 2800 3200; do
  for ps in 150 70 40; do run --config 2 --cols 23 --group 16 --long-split $t --prio-share $ps; done
done
run --config 2 --cols 23 --group 16 --long-split -1 --prio-share 40
run --config 2 --cols 23 --group 16 --long-split -1 --prio-share 20
