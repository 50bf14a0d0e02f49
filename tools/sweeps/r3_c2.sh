cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "reference_shaped or positive_gap" 2>&1 | tail -3
run() {
  timeout -k 20 200 python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>gpurun_out/r3_c2.err | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], flush=True)
" || { echo "FAILED: $*"; tail -3 gpurun_out/r3_c2.err; }
}
run --config 2
run --config 2 --autotune
run --config 2 --long-split -1
run --config 2 --long-split 800
run --config 2 --long-split 1200
run --config 2 --long-split 2000
run --config 2 --long-cols 12 --long-group 32
run --config 2 --cols 12 --group 32
run --config 2 --cols 24 --group 16
run --config 2 --max-waves 8
run --config 2 --long-helps
run --config 2 --f16 0
