#!/bin/bash
# A/B of environment settings on one workload, alternating twice: tools/ab_env_w.sh <workload> "VAR=a" "VAR=b" ...
W=$1; shift
for rep in 1 2; do
for E in "$@"; do
  printf '%-28s ' "$E"
  env $E timeout -k 10 120 python3 tools/quick_bench.py "$W" --frames 8 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("frame %.3f trace %.3f shade %.3f" % (d["frame_ms"], d["trace_ms"], d["shade_ms"]))' || exit 1
done
done
