cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
for n in 5000 20000 50000; do
run --nseq $n
run --nseq $n --no-autotune
run --nseq $n --cols 24 --group 16
run --nseq $n --cols 12 --group 32
run --nseq $n --cols 6 --group 64
run --nseq $n --cols 6 --group 64 --long-split -1
done
