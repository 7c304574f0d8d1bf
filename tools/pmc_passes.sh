#!/bin/bash
# PMC passes for the trace kernel, one rocprofv3 run per pass (counter slots:
# 8 SQ / 4 TCC / 4 TCP / 2 TA / 2 TD / 2 GRBM per pass; FETCH_SIZE takes 3 TCC, WRITE_SIZE 2).
# usage: tools/pmc_passes.sh <config> <outdir-under-gpurun_out>
set -e
CFG=${1:-c2}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${2:-pmc_$CFG}
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/gpu_prof_target.py $CFG 2 > "$OUT/$name.log" 2>&1
}
if [ -n "$TAPASS" ]; then
  run ta1 TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES
  run ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
  python3 tools/pmc_summary.py "$OUT"
  exit 0
fi
if [ -n "$QUICKMEM" ]; then  # instruction counts, lane utilisation and memory-side traffic only
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  python3 tools/pmc_summary.py "$OUT"
  exit 0
fi
if [ -n "$QUICK" ]; then
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
  python3 tools/pmc_summary.py "$OUT"
  exit 0
fi
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS
run mem1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py "$OUT"
