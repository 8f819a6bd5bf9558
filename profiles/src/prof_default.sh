#!/bin/bash
# rocprofv3 kernel stats of the default bench workload (P = 16), summarised; raw trace deleted
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_def -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_def.json 2> gpurun_out/prof_def.err
f=$(find gpurun_out/prof_def -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_def
