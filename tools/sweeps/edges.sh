cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-autotune "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'P',c['passes'],'wgs',c['workgroups'],'wq',c['work_queue'],'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'])
"
}
L="--lq 1024 --nseq 200000 --long-split -1"
run $L --cols 32 --group 32 --max-waves 4
run $L --cols 32 --group 16 --max-waves 4
run $L --cols 16 --group 16 --max-waves 4
run $L --cols 16 --group 64 --max-waves 4
run $L --cols 16 --group 32 --max-waves 4
run $L --cols 32 --group 16 --max-waves 4 --static-streams
SWG_TRACE=gpurun_out/trace_e.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-autotune $L --cols 32 --group 16 --max-waves 4 > /dev/null 2>&1; python tools/trace_timeline.py gpurun_out/trace_e.txt
