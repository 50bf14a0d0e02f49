# Geometry of the long class on config 2: columns per lane x lanes per pair, and where the class begins.
# Round 1 result: the default (64 lanes x 6 columns from 2133 pairs on) 5915-5937 GCUPS; beginning the
# class at 1200 rows 5908, at 2000 rows 5497, no long class 4787.  Forcing 32 lanes x 12 columns makes
# the planner drop the class (its longest chain would outlast the bulk) and fall back to a single
# 64 x 6 class: 4570-4790 -- those lines measure that fallback, not a 32 x 12 long class.
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
