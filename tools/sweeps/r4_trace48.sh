#!/bin/bash
# the GPU suite on the tree, then the headline's 48 launches per search one by one (rocprofv3 kernel trace): which
# (pass, segment) launches take how long, and how much of a launch is its ragged end
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4/suite2.log 2>&1 || { tail -30 gpurun_out/r4/suite2.log; exit 1; }
tail -2 gpurun_out/r4/suite2.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4/trace48 -- python3 $R/bench.py --gpus 1 --steps 3 --warmup 1 --only-headline --no-cpu-baseline --no-verify > $R/gpurun_out/r4/trace48.json 2> $R/gpurun_out/r4/trace48.err ) || { tail -5 gpurun_out/r4/trace48.err; exit 1; }
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4/trace48/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "swg_diag_dyn_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-96:]  # the last two searches
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/r4/trace48.txt", "w")
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    k = r["Kernel_Name"].split("<")[1].split(">")[0]
    print("%3d  <%s>  start %9.3f ms  dur %8.3f ms  gap %7.1f us  grid %s wg %s" % (i, k, (s - t0) / 1e6, (e - s) / 1e6, gap, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")), file=out)
out.close()
print(open("gpurun_out/r4/trace48.txt").read())
PY
