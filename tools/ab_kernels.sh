#!/bin/bash
# A/B of library builds under rocprofv3: average duration of every kernel of a quick_bench workload
#   tools/ab_kernels.sh <workload> lib1.so lib2.so ...
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  O=/tmp/abk_$$; rm -rf $O
  RT_HIP_LIBRARY="/root/repo/$L" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 /root/repo/tools/quick_bench.py "$W" --frames 30 > $O.log 2>&1 || { tail -3 $O.log; exit 1; }
  echo "== $L"
  python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print('   %-52s %4s avg %.4f ms min %.4f' % (r['Name'][:52], r['Calls'], float(r['AverageNs'])/1e6, int(r['MinNs'])/1e6))
PY
done
