cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
run() {
  # (stderr is kept and a run that prints no line says so with its exit code: round 2's copy of this helper threw
  # both away, and one of its runs could not be told apart from a hang afterwards)
  timeout -k 20 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-host-inclusive "$@" 2>gpurun_out/r2_q32.err > gpurun_out/r2_q32.out; rc=$?
  grep -q '^{' gpurun_out/r2_q32.out || { echo "$* -> NO LINE, exit code $rc"; tail -5 gpurun_out/r2_q32.err; return; }
  grep '^{' gpurun_out/r2_q32.out | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], d.get('verify',{}).get('ok'))
"
}
run --config 2 --force-bits 32
run --config 3 --force-bits 32
run --config 2 --force-bits 32 --long-split -1
run --config 2 --force-bits 32 --cols 12 --group 32
