cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 600 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
run
run --cols 32 --group 64 --max-waves 12
run --cols 16 --group 64 --max-waves 8
run --cols 16 --group 64 --max-waves 16
run --cols 32 --group 32 --max-waves 12
run --cols 32 --group 16 --max-waves 4
run --cols 28 --group 32 --max-waves 12
run --cols 20 --group 64 --max-waves 16
