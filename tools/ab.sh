#!/bin/bash
# A/B on ONE box (boxes differ by 1-2 %): alternates builds of the library, ROUNDS times.
# usage: CFGS="c2 c3" VARIANTS="base rs16 rs24" [ROUNDS=3] bash tools/ab.sh     (base = rt_amd/librtgpu.so, x = rt_amd/librtgpu_x.so)
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in $VARIANTS; do
    if [ $v = base ]; then unset RTGPU_LIB; else export RTGPU_LIB=$PWD/rt_amd/librtgpu_$v.so; fi
    timeout -k 10 200 python3 tools/gpu_perf.py $CFGS 2>/dev/null | sed -E "s/^([a-z0-9]+):.*(trace_kernel[^ ]*) kernel ms min=([0-9.]+) med=([0-9.]+).*/$v r$r \1 \2 min \3 med \4/"
  done
done
