cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'S',c.get('streams'), 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], d['kernel_ms'])
"
}
run python bench.py --steps 20 --warmup 3 --no-cpu-baseline
run python bench.py --steps 20 --warmup 3 --no-cpu-baseline
SWG_BENCH_FORCE_DIST=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline
SWG_BENCH_FORCE_DIST=1 run python bench.py --steps 20 --warmup 3 --no-cpu-baseline
run python bench.py --steps 20 --warmup 3 --no-cpu-baseline --long-split -1 --cols 12 --group 32 --max-waves 12
