# after the scheduling change: the long-class cut of config 2, model pick against autotune over query lengths,
# the int32 shapes, the multi-query batch on the big database
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
echo "== long-class cut, config 2"
STEPS=60 run --config 2
for t in 600 800 1000 1200 1400 1600 2000; do STEPS=60 run --config 2 --cols 23 --group 16 --long-split $t; done
echo "== model pick / autotune by query length (200k sequences)"
for lq in 100 200 300 430 600 800 1000 1500; do run --lq $lq --nseq 200000 --config 3; run --lq $lq --nseq 200000 --config 3 --autotune; done
echo "== int32"
STEPS=10 run --config 2 --force-bits 32
STEPS=10 run --config 3 --force-bits 32
STEPS=4 run --config 4 --force-bits 32 --nseq 300000
echo "== several queries"
timeout -k 10 300 python tools/sweeps/r2_multi_big.py 2>&1 | tail -8
