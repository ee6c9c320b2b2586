#!/bin/bash
# Per-launch durations of the shade kernels of one frame (levels in order) under rocprofv3:  tools/klevels.sh [workload] [library]
W=${1:-pool4k}; LIB=${2:-}
OUT=/root/repo/gpurun_out/klev_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export RT_HIP_LIBRARY="$LIB"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 /root/repo/tools/quick_bench.py "$W" --frames 4 > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$(ls "$OUT"/*/*kernel_trace.csv | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sh = [r for r in rows if "k_shade" in r["Kernel_Name"] or "k_fix" in r["Kernel_Name"] or "k_trace" in r["Kernel_Name"]]
n = len(sh) // 4
last = sh[-n:]          # last frame
for r in last:
    print(f'{r["Kernel_Name"].split("(")[0].replace("void rt::","")[:34]:36s} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f} us')
PY
