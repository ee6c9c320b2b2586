#!/bin/bash
# rocprofv3 kernel-trace summaries of the frame kernels (run on the GPU box):
#   tools/profile_kernels.sh [outdir]
# writes <outdir>/<workload>_kernel_stats.csv for the four workloads of tools/quick_bench.py and
# <outdir>/bench_pool4k_rocprof.json (bench.py's own line measured under the profiler).
set -u
OUT=${1:-/root/repo/gpurun_out/kstats}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (enough frames that the first launches of a process — several times longer than a steady launch — stay below 3 % of the averages)
for w in pool4k original1080 pool8k synth4k; do
  unset RT_QUEUE_SLOTS_PER_PIXEL
  F=60; [ "$w" = pool8k ] && F=30; [ "$w" = synth4k ] && F=30
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -- \
    python3 /root/repo/tools/quick_bench.py "$w" --frames $F > "$OUT/$w.log" 2>&1 || { echo "$w failed"; tail -5 "$OUT/$w.log"; exit 1; }
  cp "$(ls "$OUT/$w"/*/*kernel_stats.csv | head -1)" "$OUT/${w}_kernel_stats.csv" && echo "$w ok"
done
# the frames the roofline fraction is quoted on: RT_FLAG_BRUTE_FORCE (every surface tested for every shadow sample)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pool4k_brute" -- \
  python3 /root/repo/tools/quick_bench.py pool4k --brute --frames 60 > "$OUT/pool4k_brute.log" 2>&1 || { echo "pool4k_brute failed"; tail -5 "$OUT/pool4k_brute.log"; exit 1; }
cp "$(ls "$OUT/pool4k_brute"/*/*kernel_stats.csv | head -1)" "$OUT/pool4k_brute_kernel_stats.csv" && echo "pool4k_brute ok"
unset RT_QUEUE_SLOTS_PER_PIXEL
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- \
  python3 /root/repo/bench.py --no-cpu-baseline > "$OUT/bench_pool4k_rocprof.json" 2> "$OUT/bench.log" || { echo "bench failed"; tail -5 "$OUT/bench.log"; exit 1; }
cp "$(ls "$OUT/bench"/*/*kernel_stats.csv | head -1)" "$OUT/bench_pool4k_kernel_stats.csv" && echo "bench ok"
