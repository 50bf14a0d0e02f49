cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print(d['value'],'GCUPS step', d['ms_per_step'], d['kernel_ms'])
"
}
A="--steps 30 --warmup 3 --no-cpu-baseline"
for i in 1 2; do
echo plain; run python bench.py $A
echo dist; SWG_BENCH_FORCE_DIST=1 run python bench.py $A
done
