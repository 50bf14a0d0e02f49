cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
run
run --no-autotune
run --no-autotune --no-long-helps
run --no-autotune --long-split 1200
run --no-autotune --long-split 1400
run --no-autotune --long-split 1400 --no-long-helps
run --no-autotune --long-split 1700
run --no-autotune --long-split 2000
run --no-autotune --prio-share 100
run --no-autotune --prio-share 100000
run --config 3
run --config 1
run --nseq 20000
run --nseq 400000 --lq 200
run --config 5 --steps 5
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
