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
echo plain-pipelined; run python bench.py $A
echo plain-serial; run python bench.py $A --no-pipeline
echo dist-pipelined; SWG_BENCH_FORCE_DIST=1 run python bench.py $A
echo dist-serial; SWG_BENCH_FORCE_DIST=1 run python bench.py $A --no-pipeline
echo c3; run python bench.py $A --config 3
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
