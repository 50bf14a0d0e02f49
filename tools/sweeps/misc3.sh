cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'wq',c.get('work_queue'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
A="--steps 10 --warmup 2 --no-cpu-baseline"
for lq in 1500 2000; do echo lq $lq; run python bench.py $A --lq $lq --nseq 200000; done
echo c3; run python bench.py $A --config 3
echo c4; run python bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 4
echo c5; run python bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 5
echo c1; run python bench.py $A --config 1
