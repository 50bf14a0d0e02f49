cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'],'long',c.get('long_pairs'),'fill', d['kernel_ms']['fill'])
"
}
for w in 0 640 768 896 1024; do run --config 2 --workgroups $w; done
for w in 0 512 768; do run --config 3 --workgroups $w; done
for K in 16 20 23 24; do for w in 768 1024; do run --config 2 --uniform-len 400 --nseq 200000 --lq $((16*K)) --cols $K --group 16 --long-split -1 --workgroups $w; done; done
timeout -k 10 500 python tests/fuzz_gpu.py 300 4242 2>&1 | tail -3
