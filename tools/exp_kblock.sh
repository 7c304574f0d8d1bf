#!/bin/bash
# Occupancy experiment: builds librtgpu with other launch shapes (threads per block x
# blocks per CU x forced waves per SIMD) next to the product library and times c2/c3/c5.
#   build here:   bash tools/exp_kblock.sh build "768 1 3" "512 2 4" "640 2 5"
#   run on GPU:   bash tools/exp_kblock.sh run
# Round-1 findings: 8 / 12 / 16 waves per CU -> C2 6.66 / 5.25 / 4.61 ms,
# C3-class 5.96 / 4.60 / 3.92 ms; two blocks of 640 or 768 threads (20 / 24 waves) did not
# become co-resident (times equal to ONE such block), 2 x 512 equals 1 x 1024.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  make -C rt_amd/csrc -j8 >/dev/null
  for cfg in "$@"; do
    set -- $cfg
    ( cd rt_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math \
        -fno-slp-vectorize -I../../include -I. -DRTGPU_KBLOCK=$1 -DRTGPU_BLOCKS_PER_CU=$2 -DRTGPU_WAVES_PER_EU=$3 $4 -O2 -Rpass-analysis=kernel-resource-usage -c -o build/k_exp.o rtgpu_kernels.hip 2> build/k_exp.txt &&
      grep -A8 -E "trace_kernelILi0ELi(16ELi2|32ELi5|32ELi6)E" build/k_exp.txt | grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: *//' &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtgpu_x$1x$2w$3.so build/k_exp.o \
        $(ls build/*.o | grep -v -e k_exp.o -e rtgpu_kernels.o) -lpthread -ldl )
    echo "built rt_amd/librtgpu_x$1x$2w$3.so"
  done
else
  for lib in rt_amd/librtgpu_x*.so; do
    [ -f "$lib" ] || continue
    echo "== $lib"
    RTGPU_LIB=$PWD/$lib timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5 || exit 1
  done
  echo "== product (1024 x 1)"
  timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5
fi
