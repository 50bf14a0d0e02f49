#!/bin/bash
# Build libswg_<name>.so with extra -D flags for the kernels (A/B experiments on one GPU box).
# usage: tools/build_variant.sh <name> [-DFLAG ...]     (run after seq-align-gpu_amd/build.py)
set -e
name=$1; shift
P=/root/repo/seq-align-gpu_amd
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c -Xclang -target-feature -Xclang -load-store-opt "$@" \
  -o /tmp/swg_kernels_$name.o -x hip $P/csrc/swg_kernels.hip 2>&1 | grep -v "recognized feature" || true
objs=$(ls $P/build/*.o | grep -v swg_kernels)
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libswg_$name.so /tmp/swg_kernels_$name.o $objs -lgomp -lz -lm -ldl
echo built $P/libswg_$name.so
