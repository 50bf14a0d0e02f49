cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "diagonal_geometry" 2>&1 | tail -2
run --config 1
run --config 1 --cols 2 --group 64
run --config 1 --cols 4 --group 64
run --nseq 5000
run --nseq 20000
run --lq 64 --nseq 100000
