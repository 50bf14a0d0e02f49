cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'wq',c.get('work_queue'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
A="--steps 20 --warmup 3 --no-cpu-baseline"
echo plain; run python bench.py $A
echo dist-pipelined; SWG_BENCH_FORCE_DIST=1 run python bench.py $A
echo dist-serial; SWG_BENCH_FORCE_DIST=1 run python bench.py $A --no-pipeline
for lq in 600 800 1000 1500 2000 2500; do echo lq $lq; run python bench.py $A --lq $lq --nseq 200000; done
