#!/bin/bash
# final measurements of the round: the driver's command, the one-rank rehearsal of the N > 1 path, four gloo ranks of
# the real sharded search on the one GPU, and the driver's command under rocprofv3 (headline alone and whole)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
bash tools/sweeps/r4_bench.sh > gpurun_out/r4/final_bench.txt 2>&1 || { tail -20 gpurun_out/r4/final_bench.txt; exit 1; }
cat gpurun_out/r4/final_bench.txt
SWG_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --nseq 2000000 --steps 5 --warmup 2 > gpurun_out/r4/bench_gloo4.json 2> gpurun_out/r4/bench_gloo4.err || { tail -5 gpurun_out/r4/bench_gloo4.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4/bench_gloo4.json')); print('4 gloo ranks on one GPU:', d['value'], d['n_gpus'], d['verify']['ok'], d['per_rank']['fill_ms'])"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_headline2 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --only-headline > $R/gpurun_out/r4/prof_headline2.json 2> $R/gpurun_out/r4/prof_headline2.err ) || { tail -5 gpurun_out/r4/prof_headline2.err; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof_default2 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r4/prof_default2.json 2> $R/gpurun_out/r4/prof_default2.err ) || { tail -5 gpurun_out/r4/prof_default2.err; exit 1; }
echo profiles done
