#!/bin/bash
# round 3, final tree: kernel stats of the default bench (3 steps traced) -> gpurun_out/r03_final_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=r03_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:64]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_$tag
head -45 gpurun_out/${tag}_kernel_stats.txt
