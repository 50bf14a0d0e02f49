cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], d['kernel_ms'])
"
}
A="--steps 20 --warmup 3 --no-cpu-baseline"
echo plain; run python bench.py $A
echo dist-split; SWG_BENCH_FORCE_DIST=1 run python bench.py $A
echo plain-split-hwq1; GPU_MAX_HW_QUEUES=1 run python bench.py $A
echo plain-c3; run python bench.py $A --config 3
