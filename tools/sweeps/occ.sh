cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),'pad', c['cells_padded_over_real'], 'fill', d['kernel_ms']['fill'], 'padded-cell frac of peak', round(d['roofline']['binding_roof']['frac_of_issue_peak']*c['cells_padded_over_real'],4))
"
}
U="--no-autotune --uniform-len 400 --nseq 200000 --long-split -1"
for w in 1024 768 512; do run $U --lq 384 --cols 24 --group 16 --max-waves 4 --workgroups $w; done
for w in 768 512; do run $U --lq 512 --cols 32 --group 16 --max-waves 4 --workgroups $w; done
for w in 1024 768; do run $U --lq 256 --cols 16 --group 16 --max-waves 4 --workgroups $w; done
for w in 1024 768; do run $U --lq 320 --cols 20 --group 16 --max-waves 4 --workgroups $w; done
