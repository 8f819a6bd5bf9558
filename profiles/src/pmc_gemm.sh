#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcg_$c -- python profiles/src/pmc_gemm.py > gpurun_out/pmcg_$c.log 2>&1
  f=$(find gpurun_out/pmcg_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == sys.argv[2] and 'gemm' in r['Kernel_Name']:
        agg[r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:40] + " grid " + r.get('Grid_Size', '?')].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(sys.argv[2], k, "launches", len(v), "mean KiB", sum(v) / len(v))
PY
  rm -rf gpurun_out/pmcg_$c
done
