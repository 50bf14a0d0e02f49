# data for the cost model: 4 wavefronts per SIMD against 3 (K <= 24), and fewer passes with wider groups
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
STEPS=5 run --config 4
STEPS=5 run --config 4 --cols 32 --group 32 --max-waves 12
STEPS=5 run --config 4 --cols 24 --group 32 --max-waves 8
STEPS=5 run --config 4 --cols 24 --group 64 --max-waves 12
for lq in 550 650 750; do run --lq $lq --nseq 200000 --config 3 --max-waves 4; run --lq $lq --nseq 200000 --config 3 --max-waves 8; done
run --lq 2500 --nseq 200000 --config 3 --cols 27 --group 32 --max-waves 12
run --lq 2500 --nseq 200000 --config 3 --cols 27 --group 32 --max-waves 4
run --lq 2500 --nseq 200000 --config 3 --cols 20 --group 64 --max-waves 12
