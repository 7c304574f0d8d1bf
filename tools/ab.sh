#!/bin/bash
# A/B on one box: alternate the two builds, three rounds
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export RTGPU_LIB=$PWD/rt_amd/librtgpu_A.so; else unset RTGPU_LIB; fi
    echo "== $v round $r"; timeout -k 10 200 python3 tools/gpu_perf.py $CFGS 2>/dev/null | sed -E 's/^([a-z0-9]+):.*(trace_kernel[^ ]*) kernel ms min=([0-9.]+) med=([0-9.]+).*/\1 \2 min \3 med \4/'
  done
done
