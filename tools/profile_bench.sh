# rocprofv3 evidence for one configuration of bench.py (1 GPU): kernel trace + stats, then HBM
# counters in their own passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC sections), then SQ / LDS counters.
#   [LAUNCHES=n] bash tools/profile_bench.sh <config> [steps] [traffic key]  -> gpurun_out/prof_c<config>/{summary,traffic}.json
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
python3 $R/tools/profile_summarize.py $OUT $CFG $KEY "$ARGS" ${LAUNCHES:-1}
