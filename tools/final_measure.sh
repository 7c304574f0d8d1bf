#!/bin/bash
# Round-end measurement set for the current build, one GPU call:
#   GPU parity tests -> bench.py -> rocprofv3 kernel stats of the same bench
#   command -> FETCH_SIZE calibration -> C2 / C3 PMC passes -> perf survey ->
#   phase profile -> full-size parity against the oracle.
# usage (on the GPU box): bash tools/final_measure.sh <tag>
set -e -o pipefail
TAG=${1:-final}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 3 --cpu-spp 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
cat "$OUT/bench_under_rocprof.json"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib" -- tools/ubench/fetch_calib > "$OUT/calib.log" 2>&1
tail -2 "$OUT/calib.log"
bash tools/pmc_passes.sh c2 $TAG/pmc_c2 > "$OUT/pmc_c2.log" 2>&1
tail -12 "$OUT/pmc_c2.log"
bash tools/pmc_passes.sh c3 $TAG/pmc_c3 > "$OUT/pmc_c3.log" 2>&1
tail -8 "$OUT/pmc_c3.log"
timeout -k 10 200 python3 tools/gpu_perf.py c2 c3 c5 c5full > "$OUT/perf.log" 2>&1
cat "$OUT/perf.log"
timeout -k 10 200 python3 tools/gpu_divergence.py c2 c3 c5 > "$OUT/divergence.log" 2>&1
cat "$OUT/divergence.log"
timeout -k 10 600 python3 -u tools/gpu_full_parity.py 16 2>/dev/null | grep -v amdgpu.ids > "$OUT/full_parity.txt"
tail -1 "$OUT/full_parity.txt"
