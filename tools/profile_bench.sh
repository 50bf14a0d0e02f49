# rocprofv3 evidence for one configuration of bench.py (1 GPU): kernel trace + stats, then HBM
# counters in their own passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC sections), then SQ / LDS counters.
#   bash tools/profile_bench.sh <config> [steps] [traffic key]  -> gpurun_out/prof_c<config>/{summary,traffic}.json
# The traffic key is what bench.py looks up in profiles/traffic.json: config<C>_f16 for the packed-f16 cells (the default
# for configs 2-4), config<C>_wide for the wide int16 form, config<C>_split for both forms in one search (config 5: the f16 launches' traffic), config<C> for the int16 cells, config<C>_int32.  FETCH_SIZE is doubled: on gfx950 it reports half the
# bytes fetched, for every load shape of these kernels (tools/fetch_probe.hip, profiles/r03_fetch_size_probe.txt).
CFG=${1:-3}
STEPS=${2:-${STEPS:-20}}
KEY=${3:-config$CFG}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_c$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify $EXTRA"   # EXTRA: e.g. "--nseq 10000000" for config 4 whole
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/lds -- python3 $R/bench.py $ARGS > $OUT/lds.log 2>&1
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
summary = {"config": $CFG, "command": "python bench.py $ARGS"}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    summary["kernel_stats"] = [r for r in csv.DictReader(open(f))]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("fetch", "write", "sq", "lds"):
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in ctr.items()}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
# HBM traffic of one search: the fill kernels' FETCH_SIZE (doubled: gfx950 reports half of the fetched
# bytes, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes, mean per launch
# (the main fill only: "diag_dyn" / "fill_kernel"; a configuration's re-score leg runs swg_diag32q_kernel, listed in the summary)
fill = {k: cs for k, cs in summary["pmc_mean_per_launch"].items() if ("diag_dyn" in k or "diag_kernel" in k or "fill_kernel" in k) and "FETCH_SIZE" in cs}
if "$KEY".endswith("_split"):   # both 16-bit forms in one search: the dominant kernel's launches only (the f16 cells, FORM 2)
    fill = {k: cs for k, cs in fill.items() if k.rstrip().endswith(", 2>")}
fetch = sum(cs["FETCH_SIZE"] for cs in fill.values()); write = sum(cs.get("WRITE_SIZE", 0.0) for cs in fill.values())
json.dump({"$KEY": {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_bench.sh), mean per launch, summed over the fill kernels of one search; KiB -> bytes; FETCH_SIZE doubled (gfx950 correction of the guide's HBM section; factor 2.000 measured for these kernels' load shapes: profiles/r03_fetch_size_probe.txt)",
           "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "kernels": {k: {"FETCH_SIZE": cs["FETCH_SIZE"], "WRITE_SIZE": cs.get("WRITE_SIZE", 0.0)} for k, cs in fill.items()}}},
          open(out + "/traffic.json", "w"), indent=1)
for k, cs in summary["pmc_mean_per_launch"].items():
    if "diag" in k or "fill" in k:
        print(k, {c: "%.4g" % v for c, v in cs.items()})
for r in summary.get("kernel_stats", [])[:6]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
