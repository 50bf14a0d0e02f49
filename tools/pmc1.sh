cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --uniform-len 360 --cols 24 --group 16 --max-waves 16"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc1_$tag -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc1_$tag.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc1_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'swg_diag' in row['Kernel_Name']:
                agg[row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in sorted(agg.items()):
            print(k, 'n=%d'%len(v), 'mean=%.4g'%(sum(v)/len(v)))
PY
