#!/bin/bash
# A/B PMC comparison of two builds of the library on one workload (run on the GPU box).
#   tools/pmc_ab.sh <workload> <libA.so> <libB.so>
set -u
WL=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  export RT_HIP_LIBRARY=/root/repo/$lib
  for grp in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "b SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
             "c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
    set -- $grp; g=$1; shift
    out=/root/repo/gpurun_out/ab_${WL}_${name}_$g
    rm -rf "$out"
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out" -- python3 /root/repo/tools/quick_bench.py "$WL" --frames 2 > "$out.log" 2>&1 || { echo "pass $name $g failed"; tail -3 "$out.log"; exit 1; }
    echo "pass $name $g ok"
  done
done
