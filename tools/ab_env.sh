#!/bin/bash
# A/B of environment settings on one workload: tools/ab_env.sh <workload> "VAR=a" "VAR=b" ...
W=$1; shift
for E in "$@"; do
  printf '%-40s ' "$E"
  env $E timeout -k 10 120 python3 tools/quick_bench.py "$W" --frames 6 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("frame %.3f trace %.3f shade %.3f other %.3f" % (d["frame_ms"], d["trace_ms"], d["shade_ms"], d["other_ms"]))' || exit 1
done
