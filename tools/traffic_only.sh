# HBM traffic of one configuration's fill, the two counter passes of tools/profile_bench.sh alone (FETCH_SIZE and
# WRITE_SIZE, each in a pass of its own) -> gpurun_out/traffic_<key>/traffic.json
#   [LAUNCHES=n] [EXTRA="..."] bash tools/traffic_only.sh <config> <steps> <traffic key>
CFG=$1
STEPS=$2
KEY=$3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic_$KEY
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-host-inclusive --no-verify $EXTRA"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
python3 $R/tools/profile_summarize.py $OUT $CFG $KEY "$ARGS" ${LAUNCHES:-1}
rm -rf $OUT/fetch $OUT/write
