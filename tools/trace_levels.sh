#!/bin/bash
# per-dispatch durations of the frame kernels of one workload under rocprofv3 (GPU box):
#   tools/trace_levels.sh <workload> [tag]     (environment, e.g. RT_DEBUG_STATS, is inherited)
set -u
WL=$1; TAG=${2:-run}
OUT=/root/repo/gpurun_out/levels_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 /root/repo/tools/quick_bench.py "$WL" --frames 2 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
tail = [r for r in rows if "k_trace" in r["Kernel_Name"] or "k_shade" in r["Kernel_Name"]][-8:]
print(" ".join(f"{'T' if 'k_trace' in r['Kernel_Name'] else 'S'}{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.2f}" for r in tail))
PY
