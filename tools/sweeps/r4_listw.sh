#!/bin/bash
# the int16 list re-run of what the f16 cells flagged with a full-size workgroup (16 wavefronts beside the 98 KB
# profile instead of 4): the tests that take the route, then config 4's share with relatives (bench block 4_relatives)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flagged_pairs_rerun or relatives or wide16_range or both_16bit or golden_through_search or titin" > gpurun_out/r4/listw_tests.log 2>&1 || { tail -30 gpurun_out/r4/listw_tests.log; exit 1; }
tail -2 gpurun_out/r4/listw_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --config 7 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --no-verify 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config 7:', d['value'], d['kernel_ms'], d['config'].get('n_rescored'), d.get('first_search'), d.get('steady_state'))" || exit 1
done
