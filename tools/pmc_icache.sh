#!/bin/bash
# Instruction-fetch counters of a quick_bench workload (one extra PMC pass): tools/pmc_icache.sh [workload] [extra args]
WL=${1:-pool4k}; EXTRA=${2:-}
OUT=/root/repo/gpurun_out/pmc_icache_$WL; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/ic" -- \
  python3 /root/repo/tools/quick_bench.py "$WL" --frames 3 --host-out $EXTRA > "$OUT/ic.log" 2>&1 || { tail -5 "$OUT/ic.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
f=glob.glob(sys.argv[1]+'/ic/*/*counter_collection.csv')[0]
acc=defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:44]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'k_' not in k: continue
    req=v.get('SQC_ICACHE_REQ',0); 
    print('%-44s icache req %.3g hits %.3g misses %.3g (%.2f %%) dup %.3g  ifetch %.3g  branches %.3g  wave_cycles(x4) %.3g' % (k, req, v.get('SQC_ICACHE_HITS',0), v.get('SQC_ICACHE_MISSES',0), 100*v.get('SQC_ICACHE_MISSES',0)/max(req,1), v.get('SQC_ICACHE_MISSES_DUPLICATE',0), v.get('SQ_IFETCH',0), v.get('SQ_INSTS_BRANCH',0), v.get('SQ_WAVE_CYCLES',0)))
PY
