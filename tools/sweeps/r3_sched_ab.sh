# A/B of the machine-scheduler strategy on the round-3 kernels: build.py's -amdgpu-sched-strategy=max-ilp against
# the back end's default (libswg_defsched.so, built by hand from the same sources without the option).
cd $GRAFT_REPO_ROOT
L=seq-align-gpu_amd
cp $L/libswg.so /tmp/libswg_main.so
for lib in main defsched main defsched; do
  if [ $lib = main ]; then cp /tmp/libswg_main.so $L/libswg.so; else cp $L/libswg_defsched.so $L/libswg.so; fi
  for args in "--config 3 --steps 10" "--config 2 --steps 20" "--config 4 --steps 3" "--config 5 --steps 2"; do
    timeout -k 10 200 python bench.py $args --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib  $args:', d['value'], 'GCUPS')"
  done
done
cp /tmp/libswg_main.so $L/libswg.so
