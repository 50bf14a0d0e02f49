#!/bin/bash
# per-wavefront timeline of config 2's (and config 3's) fill on the round-4 kernels: where a 2 ms launch leaves its time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for c in 2 3; do
rm -f gpurun_out/r4/trace_c$c.txt
SWG_TRACE=gpurun_out/r4/trace_c$c.txt timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --no-verify --no-pipeline > /dev/null 2>&1 || exit 1
python tools/trace_timeline.py gpurun_out/r4/trace_c$c.txt > gpurun_out/r4/timeline_c$c.txt 2>&1 || exit 1
rm -f gpurun_out/r4/trace_c$c.txt
head -14 gpurun_out/r4/timeline_c$c.txt | cut -c1-400
done
