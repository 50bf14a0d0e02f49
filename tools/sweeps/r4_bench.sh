#!/bin/bash
# the default bench line as the driver runs it, then the one-rank rehearsal of the N > 1 path on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 560 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err || { tail -5 gpurun_out/r4/bench_default.err; exit 1; }
SWG_BENCH_FORCE_SPAWN=1 SWG_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_dist_1gpu.json 2> gpurun_out/r4/bench_dist_1gpu.err || { tail -5 gpurun_out/r4/bench_dist_1gpu.err; exit 1; }
python - <<'PY'
import json
a=json.load(open("gpurun_out/r4/bench_default.json")); b=json.load(open("gpurun_out/r4/bench_dist_1gpu.json"))
print("default  :", a["value"], a["ms_per_step"], a["config"]["workload"])
print("one rank :", b["value"], b["ms_per_step"], b["config"]["workload"])
print("ratio", a["value"]/b["value"])
for k,v in a["configs"].items(): print(k, v["value"], v["ms_per_step"], v.get("first_search"), v.get("steady_state"), (v.get("rescore") or {}).get("value"))
print(a["cpu_baseline"]["value"], a["roofline"])
PY
