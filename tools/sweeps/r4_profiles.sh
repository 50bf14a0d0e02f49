#!/bin/bash
# round-4 evidence: rocprofv3 stats of the default bench command; PMC of config 4 whole (the headline), of config 4's
# share with relatives, and of config 4's share in 2 wide passes; packing of 8 shards
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r4/prof_default.json 2> $R/gpurun_out/r4/prof_default.err ) || { tail -5 gpurun_out/r4/prof_default.err; exit 1; }
echo "default bench under rocprofv3: done"
EXTRA="--nseq 10000000" LAUNCHES=48 bash tools/profile_bench.sh 4 3 config4_whole_f16 > gpurun_out/r4/prof_c4w.log 2>&1 || { tail -5 gpurun_out/r4/prof_c4w.log; exit 1; }
mv gpurun_out/prof_c4 gpurun_out/r4/prof_c4w; echo "config 4 whole: done"
LAUNCHES=6 bash tools/profile_bench.sh 7 4 config7_f16 > gpurun_out/r4/prof_c7.log 2>&1 || { tail -5 gpurun_out/r4/prof_c7.log; exit 1; }
mv gpurun_out/prof_c7 gpurun_out/r4/prof_c7; echo "config 4 with relatives: done"
EXTRA="--group 64 --cols 32 --max-waves 12" LAUNCHES=2 bash tools/profile_bench.sh 4 4 config4_g64x32 > gpurun_out/r4/prof_c4g64.log 2>&1 || { tail -5 gpurun_out/r4/prof_c4g64.log; exit 1; }
mv gpurun_out/prof_c4 gpurun_out/r4/prof_c4g64; echo "config 4 in 2 passes: done"
timeout -k 10 300 python tools/sweeps/r4_pack_shards.py > gpurun_out/r4/pack_shards.txt 2>&1; cat gpurun_out/r4/pack_shards.txt
