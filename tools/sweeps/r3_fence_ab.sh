# A/B: the chunk fence of the profile prefetch off for every K (libswg_nofence.so: -DSWG_DYN_FENCE_ABOVE=32)
cd $GRAFT_REPO_ROOT
L=seq-align-gpu_amd
cp $L/libswg.so /tmp/libswg_main.so
for lib in main nofence main nofence; do
  if [ $lib = main ]; then cp /tmp/libswg_main.so $L/libswg.so; else cp $L/libswg_nofence.so $L/libswg.so; fi
  for args in "--config 3 --steps 10" "--config 2 --steps 20" "--config 4 --steps 3" "--config 5 --steps 2"; do
    timeout -k 10 200 python bench.py $args --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib  $args:', d['value'], 'GCUPS')"
  done
done
cp /tmp/libswg_main.so $L/libswg.so
