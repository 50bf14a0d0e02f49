#!/bin/bash
# config 4's share with relatives after the list re-run got its full-size workgroup: kernel stats + PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
LAUNCHES=6 bash tools/profile_bench.sh 7 4 config7_f16 > gpurun_out/r4/prof_c7b.log 2>&1 || { tail -5 gpurun_out/r4/prof_c7b.log; exit 1; }
rm -rf gpurun_out/r4/prof_c7b; mv gpurun_out/prof_c7 gpurun_out/r4/prof_c7b; echo "config 4 with relatives: done"
ls gpurun_out/r4/prof_c7b
