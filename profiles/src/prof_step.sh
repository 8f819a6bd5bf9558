#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step -- python bench.py --steps 1 --warmup 1 --prompts-per-gpu 1 --max-new 1024 --no-cpu-baseline > gpurun_out/prof_step.json 2> gpurun_out/prof_step.err
f=$(find gpurun_out/prof_step -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"{n:62s} calls {int(r['Calls']):6d} total_ms {int(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}%")
PY
rm -rf gpurun_out/prof_step
