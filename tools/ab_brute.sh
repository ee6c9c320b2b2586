#!/bin/bash
# A/B of library builds on pool4k, culled and RT_FLAG_BRUTE_FORCE frames, alternating (same box, same minute):
#   tools/ab_brute.sh lib1.so lib2.so ...
for rep in 1 2; do
for L in "$@"; do
  for mode in "" "--brute"; do
    printf '%-28s %-8s ' "$L" "$mode"
    RT_HIP_LIBRARY="$L" timeout -k 10 120 python3 tools/quick_bench.py pool4k --frames 8 $mode 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("frame %.3f trace %.3f shade %.3f frac %.4f" % (d["frame_ms"], d["trace_ms"], d["shade_ms"], d["shade_frac_of_peak"]))' || exit 1
  done
done
done
