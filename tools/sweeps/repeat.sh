# run-to-run spread of the default bench command
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout -k 5 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], 'GCUPS K', c['cols_per_wave'], 'long', c['long_pairs'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'])" >> gpurun_out/repeat.log || { echo FAIL >> gpurun_out/repeat.log; break; }
done
