#!/bin/bash
# Memory-instruction counts and average latencies (LDS / scalar memory / vector memory) of a quick_bench workload, two PMC
# passes: tools/pmc_latency.sh [workload] [extra args]
WL=${1:-pool4k}; EXTRA=${2:-}
OUT=/root/repo/gpurun_out/pmc_latency_$WL; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$n" -- python3 /root/repo/tools/quick_bench.py "$WL" --frames 3 --host-out $EXTRA > "$OUT/$n.log" 2>&1 || { tail -5 "$OUT/$n.log"; exit 1; }; }
run a LdsLatency SmemLatency SQ_INSTS_LDS SQ_INSTS_SMEM && run b VmemLatency SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES || exit 1
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'k_' not in k or 'init' in k: continue
    s=lambda n: sum(v.get(n,[0])); m=lambda n: (sum(v.get(n,[0]))/max(len(v.get(n,[1])),1))
    print('%-44s LDS %.3g instr, latency %.0f cyc | SMEM %.3g instr, latency %.0f | VMEM rd %.3g wr %.3g, latency %.0f | wait_any/wave_cycles %.3f' % (k, s('SQ_INSTS_LDS'), m('LdsLatency'), s('SQ_INSTS_SMEM'), m('SmemLatency'), s('SQ_INSTS_VMEM_RD'), s('SQ_INSTS_VMEM_WR'), m('VmemLatency'), s('SQ_WAIT_ANY')/max(s('SQ_WAVE_CYCLES'),1)))
PY
