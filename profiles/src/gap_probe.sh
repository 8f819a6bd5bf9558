#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 profiles/src/gap_probe.py > gpurun_out/gap_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gap -- python3 profiles/src/gap_probe.py > gpurun_out/gap_prof.log 2>&1
f=$(find gpurun_out/prof_gap -name "*kernel_stats.csv" | head -1)
echo "== un-profiled"; grep "us per launch" gpurun_out/gap_plain.log
echo "== under rocprofv3 --kernel-trace"; grep "us per launch" gpurun_out/gap_prof.log
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(r['Name'][:60], "calls", r['Calls'], "avg_us %.2f" % (float(r['AverageNs']) / 1e3))
PY
rm -rf gpurun_out/prof_gap
