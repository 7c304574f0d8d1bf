#!/bin/bash
# A/B on ONE box by an environment switch of the library (read at launch or rtgpu_create): alternates settings, ROUNDS times.
# usage: CFGS="c2 c3" VAR=RTGPU_TAIL_DRAW VALUES="- 64 16" [ROUNDS=2] bash tools/ab_env.sh     ("-" = unset)
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in $VALUES; do
    if [ "$v" = - ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 200 python3 tools/gpu_perf.py $CFGS 2>/dev/null | sed -E "s/^([a-z0-9]+):.*(trace_kernel[^ ]*) kernel ms min=([0-9.]+) med=([0-9.]+).*/$VAR=$v r$r \1 \2 min \3 med \4/"
  done
done
