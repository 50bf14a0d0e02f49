cd $GRAFT_REPO_ROOT
for args in "--uniform-len 360" "--uniform-len 360 --cols 16" "--uniform-len 360 --cols 48" "--uniform-len 360 --max-waves 6" "--uniform-len 360 --max-waves 4" "--uniform-len 360 --max-waves 3 --cols 16" "--uniform-len 360 --max-waves 1"; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$args', '->', d['value'],'GCUPS', d['config']['cols_per_wave'], d['config']['waves'], d['config']['passes'], d['config']['workgroups'], d['kernel_ms'])
"
done
