# model pick against autotune over a wider set of query lengths (200 000 sequences)
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
for lq in 50 150 250 350 450 550 650 750 900 1400 2500 4000 6000; do run --lq $lq --nseq 200000 --config 3; run --lq $lq --nseq 200000 --config 3 --autotune; done
