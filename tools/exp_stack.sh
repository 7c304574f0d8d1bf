for s in 32 16 12 8; do echo "== STACK_LDS=$s"; RTGPU_STACK_LDS=$s timeout -k 10 120 python3 tools/gpu_perf.py c3 c5 || exit 1; done
