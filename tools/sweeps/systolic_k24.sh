cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c.get('engine'), 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], d['roofline']['binding_roof']['frac_of_issue_peak'])
"
}
A="--config 2 --nseq 570000"
run $A --engine 1 --cols 24
run $A --engine 1 --cols 32
run $A --engine 1 --cols 16
run $A --engine 1 --cols 48
run $A
run --config 3 --engine 1 --cols 24
run --config 3 --engine 1 --cols 16
STEPS=3 run --config 4 --engine 1 --cols 24
STEPS=3 run --config 4 --engine 1 --cols 48
