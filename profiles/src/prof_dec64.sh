#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=${1:-8}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python3 profiles/src/decode_only.py $P ${2:-256} ${3:-bf16} > gpurun_out/dec$P.log 2>&1
grep "ms/token" gpurun_out/dec$P.log
f=$(find gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:24]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:50]
    print(f"{n:52s} calls {int(r['Calls']):6d} total_ms {int(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}%")
PY
rm -rf gpurun_out/prof_dec
