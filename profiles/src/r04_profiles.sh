#!/bin/bash
# round 4: (1) rocprofv3 kernel stats of the default bench (3 steps traced), (2) PMC passes -- FETCH_SIZE, WRITE_SIZE, MFMA busy +
# GUI_ACTIVE, each in a pass of its own with --kernel-trace only -- over the decode ring GEMVs (pmc_gemv_ring.py 128) and the training
# GEMMs (pmc_train_gemms.py).  Summaries -> gpurun_out/r04_*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=r04_default
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
echo "stats done" >> gpurun_out/r04_progress.txt
rm -f gpurun_out/r04_pmc.txt
for prog in "pmc_gemv_ring.py 128" "pmc_train_gemms.py"; do
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -d' ' -f1)_$(echo $prog | cut -d. -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc4_$t -- python3 profiles/src/$prog > gpurun_out/pmc4_$t.log 2>&1
  f=$(find gpurun_out/pmc4_$t -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmc4_$t -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$k" >> gpurun_out/r04_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52] + " grid " + r.get('Grid_Size', '?')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52] + " grid " + r.get('Grid_Size', '?')
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, c in agg.items():
    if 'gemm' not in n and 'gemv' not in n: continue
    parts = [f"{k} mean {sum(v[1:])/max(len(v)-1,1):.1f} min {min(v):.1f} max {max(v):.1f}" for k, v in c.items()]
    d = dur.get(n, [0, 0])
    print(n, "launches", len(next(iter(c.values()))), "mean_us(after first) %.1f" % (sum(d[1:]) / max(len(d) - 1, 1)), *parts)
PY
  rm -rf gpurun_out/pmc4_$t
  echo "$t done" >> gpurun_out/r04_progress.txt
done
done
cat gpurun_out/r04_pmc.txt
head -40 gpurun_out/r04_default_kernel_stats.txt
