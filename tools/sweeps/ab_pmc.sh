# A/B PMC comparison of two library builds: instruction counts and wait cycles per kernel
R=$GRAFT_REPO_ROOT
cd $R
cp seq-align-gpu_amd/libswg.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
for v in prev new; do
  if [ $v = new ]; then cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so; EXTRA="--no-long-helps"; else cp $R/seq-align-gpu_amd/libswg_$v.so $R/seq-align-gpu_amd/libswg.so; EXTRA=""; fi
  OUT=$R/gpurun_out/ab_$v; rm -rf $OUT; mkdir -p $OUT
  ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-autotune $EXTRA"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/lds -- python3 $R/bench.py $ARGS > $OUT/lds.log 2>&1
  python3 - <<PY
import csv, glob, collections
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("$v", r["Name"][:60], r["Calls"], r["AverageNs"])
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq", "lds"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in ctr.items():
    if "diag" in k:
        print("$v", k[:50], {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
done
cp /tmp/new.so $R/seq-align-gpu_amd/libswg.so
