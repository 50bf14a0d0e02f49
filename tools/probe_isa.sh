#!/bin/bash
# ISA of ONE instantiation of the lane-group kernels, in seconds instead of the minute the whole file takes:
#   tools/probe_isa.sh "32, 12" [extra -D flags]   -> /tmp/probe.s, register counts on stdout
# (swg_kernels.hip builds its variant tables from SWG_PROBE_VARIANT = "K, MAXW" alone when it is defined)
V=${1:-"32, 12"}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-sched-strategy=max-ilp \
  "-DSWG_PROBE_VARIANT=$V" "$@" -S --cuda-device-only -o /tmp/probe.s -x hip /root/repo/seq-align-gpu_amd/csrc/swg_kernels.hip 2>&1 | grep -v "recognized feature\|hip-link" 
python3 - <<'PY'
import re
t=open('/tmp/probe.s').read()
for m in re.finditer(r'\.name:\s+(_Z\d+swg_diag\w+)\n', t):
    pass
for b in re.split(r'\n\s+- \.agpr_count', t)[1:]:
    n=re.search(r'\.name:\s+(\S+)',b).group(1)
    if 'diag_dyn' in n or 'diag32q' in n:
        g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,b).group(1)
        print(n[:52], 'vgpr',g('vgpr_count'),'spill',g('vgpr_spill_count'),'sgpr',g('sgpr_count'))
PY
