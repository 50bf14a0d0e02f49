# A/B of the last pass's own geometry (option last_pass) on config 4 (3000 columns: 5 passes of 512 + one of 28 x 16 = 448)
# and on a 1100-aa and a 2300-aa query against config 3's database.
cd $GRAFT_REPO_ROOT
for lp in 0 1; do
  for args in "--config 4 --steps 4" "--config 3 --lq 1100 --steps 6" "--config 3 --lq 2300 --steps 4"; do
    timeout -k 10 200 python bench.py $args --warmup 2 --no-cpu-baseline --no-host-inclusive --last-pass $lp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('last_pass $lp  $args:', d['value'], 'GCUPS  K', c['cols_per_wave'], 'G', c['group_lanes'], 'passes', c['passes'], 'last K', c['last_pass_cols'], 'ok', d['verify']['ok'])"
  done
done
