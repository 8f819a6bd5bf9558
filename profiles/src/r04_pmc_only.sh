#!/bin/bash
# round 4, final tree: the PMC passes of r04_profiles.sh alone (FETCH_SIZE; WRITE_SIZE; MFMA busy + GUI_ACTIVE; each its own pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r04_pmc_final.txt
for prog in "pmc_gemv_ring.py 128" "pmc_train_gemms.py"; do
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -d' ' -f1)_$(echo $prog | cut -d. -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc5_$t -- python3 profiles/src/$prog > gpurun_out/pmc5_$t.log 2>&1
  f=$(find gpurun_out/pmc5_$t -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> gpurun_out/r04_pmc_final.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52] + " grid " + r.get('Grid_Size', '?')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, c in agg.items():
    if 'gemm' not in n and 'gemv' not in n: continue
    print(n, "launches", len(next(iter(c.values()))), *[f"{k} mean {sum(v[1:])/max(len(v)-1,1):.1f}" for k, v in c.items()])
PY
  rm -rf gpurun_out/pmc5_$t
done
done
cat gpurun_out/r04_pmc_final.txt
