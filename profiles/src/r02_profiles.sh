#!/bin/bash
# round 2: (1) rocprofv3 kernel stats of the default bench workload, (2) MFMA utilisation of the batched vision prefill,
# (3) HBM traffic of the decode ring GEMV (FETCH_SIZE / WRITE_SIZE in separate passes).  Summaries -> gpurun_out/r02_*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_def -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recipe-shape > gpurun_out/r02_prof_def.json 2> gpurun_out/r02_prof_def.err
f=$(find gpurun_out/prof_def -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/r02_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_def
echo "stats done" >> gpurun_out/r02_progress.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 profiles/src/pmc_prefill_batched.py > gpurun_out/r02_pmc_mfma.log 2>&1
f=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1)
k=$(find gpurun_out/pmc_mfma -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$k" > gpurun_out/r02_pmc_mfma.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[n] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    dur[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tm = tg = 0.0
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:14]:
    g = c.get('GRBM_GUI_ACTIVE', 0); m = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    util = 100 * m / (g / 8 * 1024) if g else 0
    if 'gemm' in n:
        tm += m; tg += g
    print(f"{n:42s} launches {cnt[n]:5d} dur_us {dur[n]:10.1f} MFMA_BUSY {m:14.0f} GUI_ACTIVE {g:12.0f} MfmaUtil% {util:6.1f}")
print(f"all gemm kernels, time-weighted MfmaUtil% {100 * tm / (tg / 8 * 1024):.1f}")
PY
rm -rf gpurun_out/pmc_mfma
echo "mfma done" >> gpurun_out/r02_progress.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 profiles/src/pmc_gemv_ring.py 128 > gpurun_out/r02_pmc_$c.log 2>&1
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c >> gpurun_out/r02_pmc_gemv.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemv' in r.get('Kernel_Name', '') and r['Counter_Name'] == sys.argv[2]:
        by[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:50]].append(float(r['Counter_Value']))
for n, v in by.items():
    print(sys.argv[2], n, "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
PY
  rm -rf gpurun_out/pmc_$c
done
echo "pmc done" >> gpurun_out/r02_progress.txt
