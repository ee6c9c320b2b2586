#!/bin/bash
# PMC passes for the frame kernels (run on the GPU box).  One counter group per rocprofv3 run
# (rocprofv3 does not split counters over passes; gfx950 limits: 8 SQ, 4 TCC, 2 GRBM per run).
#   tools/profile_pmc.sh [workload] [outdir] [extra quick_bench.py arguments, e.g. --brute]
set -u
WL=${1:-pool4k}
OUT=${2:-/root/repo/gpurun_out/pmc_$WL}
EXTRA=${3:-}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 /root/repo/tools/quick_bench.py "$WL" --frames 3 --host-out $EXTRA > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU &&
run fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run write WRITE_SIZE GRBM_COUNT &&
# LDS: instructions, busy and stalled cycles (bank conflicts of the scene staging / the kernels' LDS tables)
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS
