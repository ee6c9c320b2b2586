#!/bin/bash
# A/B of environment settings on pool4k (culled + brute), alternating: tools/ab_env2.sh "VAR=a" "VAR=b" ...
for rep in 1 2; do
for E in "$@"; do
  for mode in "" "--brute"; do
    printf '%-28s %-8s ' "$E" "$mode"
    env $E timeout -k 10 120 python3 tools/quick_bench.py pool4k --frames 8 $mode 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("frame %.3f trace %.3f shade %.3f frac %.4f" % (d["frame_ms"], d["trace_ms"], d["shade_ms"], d["shade_frac_of_peak"]))' || exit 1
  done
done
done
