#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4/check6.txt
: > $O
for o in "" "f16=0" "engine=1 cols_per_wave=16" "engine=1 cols_per_wave=24"; do echo "== peptides $o" >> $O; timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 $o 2>&1 | grep "lq\|rror" >> $O || { cat $O; exit 1; }; done
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r4/suite7.log 2>&1; tail -3 gpurun_out/r4/suite7.log >> $O
grep -q " passed" gpurun_out/r4/suite7.log && ! grep -q " failed" gpurun_out/r4/suite7.log || { tail -40 gpurun_out/r4/suite7.log; exit 1; }
timeout -k 10 400 python tests/fuzz_gpu.py 300 45 > gpurun_out/r4/fuzz4.log 2>&1; tail -1 gpurun_out/r4/fuzz4.log >> $O
cat $O
