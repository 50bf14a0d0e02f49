cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'ovl',c.get('classes_overlapped'),'fill', d['kernel_ms']['fill'])
"
}
echo "== lengths clamped at 2400: default plan, then the bulk's own geometry as the long class"
run --config 2 --max-len 2400
for t in 800 1000 1200 1400; do run --config 2 --max-len 2400 --cols 23 --group 16 --long-group 16 --long-split $t; done
run --config 2 --max-len 2400 --cols 23 --group 16 --long-split -1
echo "== full config 2 for reference"
run --config 2
for t in 1000 1200; do run --config 2 --cols 23 --group 16 --long-group 16 --long-split $t; done
