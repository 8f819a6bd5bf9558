#!/bin/bash
# rocprofv3 kernel stats of the default bench workload (3 steps traced: warm-up + timed + the instrumented phase-split step)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-step}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_def -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recipe-shape > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
f=$(find gpurun_out/prof_def -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${TAG}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_def
tail -c 1500 gpurun_out/${TAG}_bench.json
head -30 gpurun_out/${TAG}_kernel_stats.txt
