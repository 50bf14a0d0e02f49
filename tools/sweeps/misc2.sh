cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'wq',c.get('work_queue'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
A="--steps 30 --warmup 3 --no-cpu-baseline --no-autotune"
echo plain; run python bench.py $A
for q in 4 8 16; do echo dist hwq $q; GPU_MAX_HW_QUEUES=$q SWG_BENCH_FORCE_DIST=1 run python bench.py $A; done
echo dist hwq 8 serial; GPU_MAX_HW_QUEUES=8 SWG_BENCH_FORCE_DIST=1 run python bench.py $A --no-pipeline
echo plain hwq 8; GPU_MAX_HW_QUEUES=8 run python bench.py $A
