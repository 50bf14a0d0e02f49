cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], 'step', d['ms_per_step'])
"
}
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run
run --long-split -1
run --cols 12 --group 32 --max-waves 12
run --cols 12 --group 32 --max-waves 12 --long-split 1000
run --cols 12 --group 32 --max-waves 12 --long-split 2000
run --cols 12 --group 32 --max-waves 16
run --cols 24 --group 16 --max-waves 16
run --cols 24 --group 16 --max-waves 16 --long-split 800
run --cols 24 --group 16 --max-waves 16 --long-split 500
run --cols 24 --group 16 --max-waves 8
run --config 3
