# Geometry of the long class on config 2: columns per lane x lanes per pair, and where the class begins.
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 200 python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-host-inclusive "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'), 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'], flush=True)
"
}
run
run --long-cols 12 --long-group 32
run --long-cols 6 --long-group 64
run --long-cols 12 --long-group 32 --long-split 2000
run --long-cols 12 --long-group 32 --long-split 1200
run --long-cols 23 --long-group 16 --long-split 1500
run --long-split 2000
run --long-split 1200
run --long-split -1
run
