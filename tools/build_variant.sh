#!/bin/bash
# Build libswg_<name>.so with extra -D flags for the kernels AND the host layer that shares
# csrc/swg_internal.h with them (A/B experiments on one GPU box).
# usage: tools/build_variant.sh <name> [-DFLAG ...]     (run after seq-align-gpu_amd/build.py)
set -e
name=$1; shift
P=/root/repo/seq-align-gpu_amd
for src in swg_kernels.hip swg_api.cpp; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
    -o /tmp/${src}_$name.o -x hip $P/csrc/$src 2>&1 | grep -v "recognized feature" || true
done
objs=$(ls $P/build/*.o | grep -v "swg_kernels\|swg_api")
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libswg_$name.so /tmp/swg_kernels.hip_$name.o /tmp/swg_api.cpp_$name.o $objs -lgomp -lz -lm -ldl
echo built $P/libswg_$name.so
