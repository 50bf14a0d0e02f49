cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 600 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
for c in 2 3; do run --config $c --no-autotune; run --config $c; done
STEPS=3 run --config 4 --no-autotune
STEPS=3 run --config 5 --no-autotune
for lq in 100 200 300 430 600 1000 1500 2000 2500; do run --lq $lq --nseq 200000 --no-autotune; done
