cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r2_bench_default2.json 2> gpurun_out/r2_bench_default2.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_default2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench_default2.json'))
for k,b in list(d["configs"].items())+[("4full",d["scaling_reference"])]:
    print(k, b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms"], b["roofline"]["binding_roof"]["frac_of_issue_peak"], b["config"]["cols_per_wave"], b["config"]["group_lanes"], b["config"]["passes"], b["config"]["long_pairs"], b["config"]["classes_overlapped"], b.get("verify",{}).get("ok"))
print(d["host_inclusive"]); print(d["cpu_baseline"])
PY
timeout -k 10 300 python bench.py --config 2 --force-bits 32 --cols 12 --group 32 --steps 5 --no-cpu-baseline 2>&1 | tail -5 | cut -c1-600
