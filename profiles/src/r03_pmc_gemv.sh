#!/bin/bash
# round 3: HBM traffic of the decode ring GEMVs at 128 rows, re-measured on the round's tree (FETCH_SIZE / WRITE_SIZE in separate
# rocprofv3 --pmc passes, as MI355X_MICROARCH.md prescribes) + the launches' durations from the kernel trace of the same passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r03_pmc_gemv.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 profiles/src/pmc_gemv_ring.py 128 > gpurun_out/r03_pmc_$c.log 2>&1
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmc_$c -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$k" $c >> gpurun_out/r03_pmc_gemv.txt <<'PY'
import csv, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemv' in r.get('Kernel_Name', '') and r['Counter_Name'] == sys.argv[3]:
        by[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:50]].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if 'gemv' in r.get('Kernel_Name', ''):
        dur[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:50]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in by.items():
    d = dur.get(n, [0.0])
    print(sys.argv[3], n, "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v), "mean_us %.2f" % (sum(d) / len(d)))
PY
  rm -rf gpurun_out/pmc_$c
done
cat gpurun_out/r03_pmc_gemv.txt
