"""Summary of one tools/profile_bench.sh run: kernel stats + mean PMC counters per kernel (summary.json) and the HBM
traffic per launch of the fill (traffic.json, the entry bench.py looks up in profiles/traffic.json).
usage: python3 tools/profile_summarize.py <out dir> <config> <traffic key> "<bench.py arguments>" [launches per search]"""
import csv, glob, json, collections, os, sys
KEY = sys.argv[3]
out = sys.argv[1]
summary = {"config": int(sys.argv[2]), "command": "python bench.py " + sys.argv[4]}
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
# KERNEL_SUFFIX=", 1>" (environment): only the fill kernels whose name ends so -- a run whose legs use different cell forms
# (bench.py --config 6: the wide form, then plain int16) -- and their mean per dispatch, as for the split keys
SUFFIX = os.environ.get("KERNEL_SUFFIX", "")
PER_DISPATCH = KEY.endswith("_split") or KEY.endswith("_split16") or bool(SUFFIX)
if PER_DISPATCH:   # both 16-bit forms in one search: the dominant kernel's launches only (the f16 cells, FORM 2)
    fill = {k: cs for k, cs in fill.items() if k.rstrip().endswith(SUFFIX or ", 2>")}
# per search: every fill dispatch's counters added up, divided by the searches the run made (one swg_zero2_kernel each);
# per launch: that divided by LAUNCHES = bench.py's roofline.launches_per_step (passes x segments; 1 for a single pass,
# whose two classes run side by side as one step of the fill)
searches = max(1, len(ctr.get("swg_zero2_kernel", {}).get("FETCH_SIZE", [])))
launches = max(1, int(sys.argv[5]) if len(sys.argv) > 5 else 1)
fetch = sum(sum(ctr[k]["FETCH_SIZE"]) for k in fill) / searches / launches
write = sum(sum(ctr[k].get("WRITE_SIZE", [0.0])) for k in fill) / searches / launches
if PER_DISPATCH:   # (only some of the run's searches -- not its re-score leg's -- launch this kernel: its mean per dispatch)
    fetch = sum(cs["FETCH_SIZE"] for cs in fill.values())
    write = sum(cs.get("WRITE_SIZE", 0.0) for cs in fill.values())
HOW_PD = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic_only.sh / profile_bench.sh): the MEAN PER DISPATCH of "
          "the fill kernel whose name ends '%s' (the run's legs use several cell forms; a launch of bench.py's roofline is one such dispatch); "
          "KiB -> bytes; FETCH_SIZE doubled (gfx950 correction of the guide's HBM section; factor 2.000 measured for these kernels' load "
          "shapes: profiles/r03_fetch_size_probe.txt)" % (SUFFIX or ", 2>"))
json.dump({KEY: {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
           "how": HOW_PD if PER_DISPATCH else "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_bench.sh): the counters of every fill dispatch of the run added up, divided by its searches and by the launches of one search's fill (%d here: bench.py's roofline.launches_per_step); KiB -> bytes; FETCH_SIZE doubled (gfx950 correction of the guide's HBM section; factor 2.000 measured for these kernels' load shapes: profiles/r03_fetch_size_probe.txt)" % launches,
           "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "searches": searches, "launches_per_search": launches,
           "kernels": {k: {"FETCH_SIZE": cs["FETCH_SIZE"], "WRITE_SIZE": cs.get("WRITE_SIZE", 0.0)} for k, cs in fill.items()}}},
          open(out + "/traffic.json", "w"), indent=1)
for k, cs in summary["pmc_mean_per_launch"].items():
    if "diag" in k or "fill" in k:
        print(k, {c: "%.4g" % v for c, v in cs.items()})
for r in summary.get("kernel_stats", [])[:6]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
