#!/bin/bash
# Two work items per lane (kernel mode 8, RTGPU_DUAL=1): builds librtgpu with other thresholds
# (round ends below R traversing lanes, second-slot pass from P waiting results up, exchange from S lanes up)
#   build here:   bash tools/exp_dual.sh build "48 32 8" "40 32 8"
#   run on GPU:   bash tools/exp_dual.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  make -C rt_amd/csrc -j8 >/dev/null
  for cfg in "$@"; do
    set -- $cfg
    ( cd rt_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math \
        -fno-slp-vectorize -I../../include -I. -DRTGPU_DUAL_REFILL=$1 -DRTGPU_DUAL_PASS_MIN=$2 -DRTGPU_DUAL_SWAP_MIN=$3 $4 -O2 -c -o /tmp/k_dual_$1_$2_$3.o rtgpu_kernels.hip &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtgpu_d$1p$2s$3.so /tmp/k_dual_$1_$2_$3.o \
        $(ls build/*.o | grep -v -e k_exp.o -e rtgpu_kernels.o) -lpthread -ldl
      echo "built rt_amd/librtgpu_d$1p$2s$3.so" ) &
  done
  wait
else
  for lib in rt_amd/librtgpu_d*.so; do
    [ -f "$lib" ] || continue
    echo "== $lib"
    RTGPU_DUAL=1 RTGPU_LIB=$PWD/$lib timeout -k 10 120 python3 tools/gpu_perf.py ${CFGS:-c2 c4} 2>&1 | grep -o "^c[0-9a-z]*:\|trace_kernel.*Mrays/s" || exit 1
  done
  echo "== product (mode 7)"
  timeout -k 10 120 python3 tools/gpu_perf.py ${CFGS:-c2 c4} 2>&1 | grep -o "^c[0-9a-z]*:\|trace_kernel.*Mrays/s"
fi
