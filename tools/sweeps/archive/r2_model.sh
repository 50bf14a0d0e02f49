# (1) overhead fit: uniform database, one lane-group width, several K -> fill ms per (K, G)
# (2) config 2: long-class alternatives
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], 'res', c['residues_this_gpu'])
"
}
echo "== uniform 400, G16"
for K in 4 8 12 16 20 24 28 32; do run --config 2 --uniform-len 400 --nseq 200000 --lq $((16*K)) --cols $K --group 16 --long-split -1; done
echo "== uniform 400, G32"
for K in 4 8 16 24; do run --config 2 --uniform-len 400 --nseq 200000 --lq $((32*K)) --cols $K --group 32 --long-split -1; done
echo "== uniform 400, G64"
for K in 4 8 16; do run --config 2 --uniform-len 400 --nseq 200000 --lq $((64*K)) --cols $K --group 64 --long-split -1; done
echo "== config 2 long-class alternatives"
STEPS=20
run --config 2
run --config 2 --long-split -1
for t in 1000 1500 2000 2500 3000; do run --config 2 --long-split $t; done
run --config 2 --long-group 32 --long-cols 12
run --config 2 --long-group 32 --long-cols 12 --long-split 2000
run --config 2 --long-cols 8
run --config 2 --long-cols 12
run --config 2 --long-helps
