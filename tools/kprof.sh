#!/bin/bash
# Per-kernel average durations of a quick_bench workload under rocprofv3 (run on the GPU box):
#   tools/kprof.sh [workload] [frames]
W=${1:-pool4k}; F=${2:-6}
OUT=/root/repo/gpurun_out/kprof_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 /root/repo/tools/quick_bench.py "$W" --frames "$F" > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$(ls "$OUT"/*/*kernel_stats.csv | head -1)" "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2])
for r in rows:
    name = r["Name"].split("(")[0].replace("void rt::", "")
    print(f"{name:45s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.1f} us  per-frame {float(r['TotalDurationNs'])/frames/1e6:7.3f} ms")
PY
tail -1 "$OUT.log" | cut -c1-200
