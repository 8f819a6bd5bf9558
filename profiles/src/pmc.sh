#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python profiles/src/pmc_gemv.py ${1:-16} > gpurun_out/pmc_$c.log 2>&1
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemv' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == sys.argv[2]]
print(sys.argv[2], "launches", len(vals), "mean", sum(vals) / max(1, len(vals)), "min", min(vals), "max", max(vals))
PY
  rm -rf gpurun_out/pmc_$c
done
