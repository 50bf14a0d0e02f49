cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'wgs',c['workgroups'],'long',c.get('long_pairs'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
for w in 0 704 640 576 512; do run --no-autotune --workgroups $w; done
for w in 0 896 768 640; do run --no-autotune --config 3 --cols 16 --group 32 --workgroups $w; done
for w in 0 896 768; do run --no-autotune --lq 200 --nseq 200000 --workgroups $w; done
