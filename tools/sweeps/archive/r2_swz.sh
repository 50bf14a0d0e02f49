# LDS swizzle A/B: correctness of the swizzled build, throughput and LDS conflict counters of both
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or geometry or random_database or multipass or many_queries or wide16 or high_similarity" 2>&1 | tail -4 || exit 1
VARIANTS="noswz new" STEPS=30 EXTRA="--config 2" bash tools/sweeps/ab.sh
cp seq-align-gpu_amd/libswg.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
for v in noswz new; do
  if [ $v = new ]; then cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so; else cp $R/seq-align-gpu_amd/libswg_$v.so $R/seq-align-gpu_amd/libswg.so; fi
  for cfg in 2 3; do
    OUT=$R/gpurun_out/swz_${v}_c$cfg; rm -rf $OUT; mkdir -p $OUT
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 $R/bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify > $OUT.log 2>&1
    python3 - <<PY
import csv, glob, collections
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ctr[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in ctr.items():
    if "diag" in k:
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        print("$v config $cfg", k[:44], "conflict/active %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"])), {c: "%.4g" % x for c, x in m.items()})
PY
  done
done
cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so
