for v in 0 1; do echo "== NO_NODE_CACHE=$v"; RTGPU_NO_NODE_CACHE=$v timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5 || exit 1; done
for r in 0 16; do echo "== REFILL=$r"; RTGPU_REFILL=$r timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5 || exit 1; done
for c in 1 2 4; do echo "== CHUNK=$c"; RT_CHUNK=$c timeout -k 10 120 python3 tools/gpu_perf.py c2 c3 c5 || exit 1; done
