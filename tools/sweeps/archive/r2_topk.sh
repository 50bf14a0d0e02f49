cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "topk or golden_through_search or searches_in_flight" 2>&1 | tail -3 || exit 1
cp seq-align-gpu_amd/libswg.so /tmp/new.so
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in prev new; do
  if [ $v = new ]; then cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so; else cp $R/seq-align-gpu_amd/libswg_$v.so $R/seq-align-gpu_amd/libswg.so; fi
  for cfg in 2 3; do
    OUT=$R/gpurun_out/topk_${v}_c$cfg; rm -rf $OUT; mkdir -p $OUT
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive --no-verify > $OUT.log 2>&1
    grep '^{' $OUT.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v config $cfg', d['value'], 'step', d['ms_per_step'], 'fill', d['kernel_ms']['fill'], 'search_total', d['kernel_ms']['search_total'])"
    python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "topk" in r["Name"] or "diag" in r["Name"]:
            print("   ", r["Name"][:52], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so
