#!/bin/bash
# round 4, run 17 (the tree with the 129-256-row ring GEMV, default P = 22): PMC passes over the decode GEMVs at 176 and 128 rows
# (FETCH_SIZE; WRITE_SIZE; MFMA busy + GUI_ACTIVE; each its own pass), then rocprofv3 kernel stats of the default bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_pmc_p22.txt
for rows in 176 128; do
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -d' ' -f1)_$rows
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc6_$t -- python3 profiles/src/pmc_gemv_ring.py $rows > gpurun_out/pmc6_$t.log 2>&1
  f=$(find gpurun_out/pmc6_$t -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmc6_$t -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$k" $rows >> gpurun_out/r04_pmc_p22.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52]
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, c in agg.items():
    if 'gemv' not in n: continue
    d = dur.get(n, [0])
    print(sys.argv[3], "rows", n, "launches", len(next(iter(c.values()))), "mean_us %.2f" % (sum(d[1:]) / max(len(d) - 1, 1)),
          *[f"{k} mean {sum(v[1:])/max(len(v)-1,1):.1f}" for k, v in c.items()])
PY
  rm -rf gpurun_out/pmc6_$t
done
done
cat gpurun_out/r04_pmc_p22.txt
tag=r04_default_p22
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:48]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:64]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_$tag
head -30 gpurun_out/${tag}_kernel_stats.txt
cut -c1-300 gpurun_out/${tag}_bench.json
