cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
B="--no-autotune --cols 24 --group 16 --max-waves 4 --long-split 1400"
run $B --no-pipeline
run $B --steps 1 --warmup 0
run $B --steps 2 --warmup 0
run $B --steps 5 --warmup 0
SWG_TRACE=gpurun_out/trace_dyn5.txt python bench.py --steps 5 --warmup 0 --no-cpu-baseline $B 2>/dev/null | grep '^{' | cut -c1-400
python tools/trace_timeline.py gpurun_out/trace_dyn5.txt
