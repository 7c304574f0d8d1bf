#!/bin/bash
# FETCH_SIZE calibration of the access patterns this kernel has: writes <out>/fetch_calibration.txt
# usage (on the GPU box): bash tools/fetch_calibration.sh <outdir-under-gpurun_out>
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-calib}
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib" -- tools/ubench/fetch_calib > "$OUT/calib_stdout.txt" 2> "$OUT/calib.log"
python3 tools/fetch_calibration.py "$OUT/calib" "$OUT/calib_stdout.txt" > "$OUT/fetch_calibration.txt"
cat "$OUT/fetch_calibration.txt"
