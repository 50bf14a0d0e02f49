cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-host-inclusive "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', d['dtype'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'resc',c['n_rescored'],'fill', d['kernel_ms']['fill'], 'rescore', d['kernel_ms']['rescore'], d.get('verify',{}).get('ok'))
"
}
run --config 4 --nseq 200000 --force-bits 32
run --config 5 --force-bits 32
run --config 3 --lq 1500 --nseq 100000 --force-bits 32
