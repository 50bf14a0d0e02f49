cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --cols 24 --group 16 --max-waves 4 --long-cols 12 --long-group 32 --long-split 1000"
python3 $R/bench.py $ARGS 2>/dev/null | grep '^{' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'])"
rm -rf $R/gpurun_out/tr21
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr21 -- python3 $R/bench.py $ARGS > $R/gpurun_out/tr21.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/tr21/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    t0=int(rows[0]['Start_Timestamp'])
    for r in rows[-40:]:
        print(r['Kernel_Name'][:40].ljust(40), r['Stream_Id'], r['Queue_Id'], (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r['Grid_Size_X'], r['Workgroup_Size_X'])
PY
