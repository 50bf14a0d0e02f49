cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 600 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),'wq',c.get('work_queue'),'pad', c['cells_padded_over_real'], 'resc', c['n_rescored'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
A="--steps 10 --warmup 2 --no-cpu-baseline"
for lq in 2500 3000; do echo lq $lq; run python bench.py $A --lq $lq --nseq 200000; done
echo c4; run python bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 4
echo c4 static; run python bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 4 --engine 2 --static-streams
echo c5; run python bench.py --steps 3 --warmup 1 --no-cpu-baseline --config 5
echo c2; run python bench.py $A
