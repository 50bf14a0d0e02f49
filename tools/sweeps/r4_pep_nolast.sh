#!/bin/bash
# upper bound for everything the last-row path of the work-queue kernel costs on peptides: the same kernel (K=8 only)
# with that path compiled out (wrong scores: timing only) against the same build with it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
cp seq-align-gpu_amd/libswg.so /tmp/new.so
for v in k8 nolast; do
  cp seq-align-gpu_amd/libswg_$v.so seq-align-gpu_amd/libswg.so
  echo "== $v"
  timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 batch_blocks=64 2>&1 | grep "lq\|Error\|error" 
done
cp /tmp/new.so seq-align-gpu_amd/libswg.so
