# A/B of library builds on one box: VARIANTS="prev new" [EXTRA="bench flags"] bash tools/sweeps/ab.sh
# (libswg_<name>.so next to libswg.so; "new" is the built library)
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 20 400 python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', d['value'],'GCUPS', c['engine'], 'K',c['cols_per_wave'],'G',c.get('group_lanes'),'W',c['waves'],'wgs',c['workgroups'], 'long',c.get('long_pairs'),c.get('long_cols_per_lane'),c.get('long_streams'),'pad', c['cells_padded_over_real'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])
"
}
cp seq-align-gpu_amd/libswg.so /tmp/new.so
for v in ${VARIANTS}; do
  if [ $v = new ]; then cp /tmp/new.so seq-align-gpu_amd/libswg.so; else cp seq-align-gpu_amd/libswg_$v.so seq-align-gpu_amd/libswg.so; fi
  echo "== $v"; run --no-autotune ${EXTRA}; run --no-autotune ${EXTRA}; run --no-autotune --config 3
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
