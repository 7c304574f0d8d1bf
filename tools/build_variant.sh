#!/bin/bash
# Builds rt_amd/librtgpu_<name>.so with extra -D flags for the kernel file (A/B runs with tools/ab.sh).
# usage: bash tools/build_variant.sh <name> -DRTGPU_REFILL_SMALL=24 ...
set -e
cd "$(dirname "$0")/../rt_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. \
  -fno-slp-vectorize -O2 "$@" -c -o build/k_$name.o rtgpu_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtgpu_$name.so build/k_$name.o \
  $(ls build/*.o | grep -v -e '/k_' -e rtgpu_kernels.o) -lpthread -ldl
echo "built rt_amd/librtgpu_$name.so"
