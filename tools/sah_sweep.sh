#!/bin/bash
# SAH parameter sweep (experiment): kernel time of c2/c3/c5 per RTGPU_SAH setting
for sah in "1,1,4" "1,2,4" "1,3,4" "1,1,2" "1,1,1" "1,2,2" "2,1,4"; do
  echo "== RTGPU_SAH=$sah"
  RTGPU_SAH=$sah timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5 || exit 1
done
