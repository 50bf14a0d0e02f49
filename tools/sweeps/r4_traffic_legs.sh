#!/bin/bash
# HBM counters of the bench legs that had no measurement of their own: config 5's rescore leg (f16 + plain int16),
# the stress variant on the wide form and on plain int16, the peptides on the systolic engine
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
EXTRA="--wide16 0" LAUNCHES=8 bash tools/traffic_only.sh 5 2 config5_split16 > gpurun_out/r4/traffic_c5s16.log 2>&1 || { tail -5 gpurun_out/r4/traffic_c5s16.log; exit 1; }
echo "config 5, wide16 = 0: done"
KERNEL_SUFFIX=", 1>" LAUNCHES=4 bash tools/traffic_only.sh 6 3 config6_wide > gpurun_out/r4/traffic_c6w.log 2>&1 || { tail -5 gpurun_out/r4/traffic_c6w.log; exit 1; }
echo "stress, wide: done"
KERNEL_SUFFIX=", 0>" EXTRA="--wide16 0" LAUNCHES=4 bash tools/traffic_only.sh 6 2 config6 > gpurun_out/r4/traffic_c6.log 2>&1 || { tail -5 gpurun_out/r4/traffic_c6.log; exit 1; }
echo "stress, int16 + int32: done"
LAUNCHES=1 bash tools/traffic_only.sh 8 20 config8_systolic_f16 > gpurun_out/r4/traffic_c8.log 2>&1 || { tail -5 gpurun_out/r4/traffic_c8.log; exit 1; }
echo "peptides: done"
cat gpurun_out/traffic_*/traffic.json | grep -E '^ "|hbm_bytes|searches|launches_per_search'
