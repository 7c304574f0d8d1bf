#!/bin/bash
# Round-end measurement set for the current build, one GPU call:
#   GPU parity tests -> ubenches (VALU issue, HBM stream, vector-memory records, FETCH_SIZE calibration) -> PMC passes of C2,
#   C3 as stated, C4's frame and C5's frame -> profiles/traffic.json from them -> bench.py -> rocprofv3 kernel stats of the same
#   bench command -> the LDS pass and the two-rays-per-lane experiment on C2 -> perf survey -> phase profile -> host path.
# usage (on the GPU box): bash tools/final_measure.sh <tag>;  then here: python tools/update_profiles.py <tag> <dest>
set -e -o pipefail
TAG=${1:-final}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 120 ./tools/ubench/valu_issue > "$OUT/valu_issue.txt" 2>&1
timeout -k 10 120 ./tools/ubench/hbm_stream > "$OUT/hbm_stream.txt" 2>&1
timeout -k 10 200 ./tools/ubench/node_fetch > "$OUT/node_fetch.txt" 2>&1
tail -2 "$OUT/node_fetch.txt"
bash tools/fetch_calibration.sh $TAG > "$OUT/calib_run.log" 2>&1
cat "$OUT/fetch_calibration.txt"
bash tools/pmc_passes.sh c2 $TAG/pmc_c2 > "$OUT/pmc_c2.log" 2>&1
tail -8 "$OUT/pmc_c2.log"
bash tools/pmc_passes.sh c3full $TAG/pmc_c3full > "$OUT/pmc_c3full.log" 2>&1
tail -8 "$OUT/pmc_c3full.log"
QUICKMEM=1 bash tools/pmc_passes.sh c4 $TAG/pmc_c4 > "$OUT/pmc_c4.log" 2>&1
tail -4 "$OUT/pmc_c4.log"
bash tools/pmc_passes.sh c5full16 $TAG/pmc_c5full16 > "$OUT/pmc_c5full16.log" 2>&1
tail -8 "$OUT/pmc_c5full16.log"
# bench.py reads the PMC figures of ITS kernels from profiles/traffic.json: refresh it from the passes above before the bench
# runs (the same command is run again on the build machine after the call, from the merged files, with the same result)
python3 tools/update_profiles.py $TAG ${DEST:-r03} > "$OUT/update_profiles.log" 2>&1
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-400 "$OUT/bench.json"
# the timed C2 region alone (every launch of the headline kernel in this trace is a C2 frame), then the whole default run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --cpu-spp 0 --no-also > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
cut -c1-200 "$OUT/bench_under_rocprof.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_also" -- python3 bench.py --cpu-spp 0 > "$OUT/bench_under_rocprof_also.json" 2> "$OUT/stats_also.log"
LDSPASS=1 bash tools/pmc_passes.sh c2 $TAG/pmc_c2_lds > "$OUT/pmc_c2_lds.log" 2>&1
RTGPU_RAYS2=1 QUICK=1 bash tools/pmc_passes.sh c2 $TAG/pmc_c2_rays2 > "$OUT/pmc_c2_rays2.log" 2>&1
tail -4 "$OUT/pmc_c2_rays2.log"
timeout -k 10 300 python3 tools/gpu_perf.py c2 c3 c5 c5full c3full c5full16 c4 c2lens 2>/dev/null > "$OUT/perf.log"
RTGPU_RAYS2=1 timeout -k 10 300 python3 tools/gpu_perf.py c2 c4 2>/dev/null > "$OUT/perf_rays2.log"
cat "$OUT/perf.log"
timeout -k 10 200 python3 tools/gpu_divergence.py c2 c3 c5 2>/dev/null > "$OUT/divergence.log"
cat "$OUT/divergence.log"
timeout -k 10 100 python3 tools/gpu_host_path.py 2>/dev/null > "$OUT/host_path.txt"
cat "$OUT/host_path.txt"
