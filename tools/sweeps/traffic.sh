# HBM traffic of one config-2 search without autotune trials in the way: FETCH_SIZE / WRITE_SIZE per kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in default nohelps; do
  EXTRA=""; [ $tag = nohelps ] && EXTRA="--no-long-helps"
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$R/gpurun_out/traffic_${tag}_$c; rm -rf $OUT
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-autotune $EXTRA > /dev/null 2>&1
  done
  python3 - <<PY
import csv, glob, collections
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/traffic_${tag}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for k, cs in ctr.items():
    if "diag" in k:
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]); w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        print("$tag", k, "launches", len(cs["FETCH_SIZE"]), "FETCH KiB %.0f WRITE KiB %.0f -> %.1f MB" % (f, w, (2 * f + w) * 1024 / 1e6))
        tot += (2 * f + w) * 1024
print("$tag total %.1f MB per search" % (tot / 1e6))
PY
done
