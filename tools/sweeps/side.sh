cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'long',c.get('long_pairs'),'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'], 'total', d['kernel_ms']['search_total'])
"
}
A="--steps 30 --warmup 4 --no-cpu-baseline --no-autotune"
for dp in 2 3; do echo dist depth $dp; SWG_BENCH_FORCE_DIST=1 run python bench.py $A --depth $dp; SWG_BENCH_FORCE_DIST=1 run python bench.py $A --depth $dp; done
for dp in 2 3; do echo plain depth $dp; run python bench.py $A --depth $dp; run python bench.py $A --depth $dp; done
