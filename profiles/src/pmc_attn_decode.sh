#!/bin/bash
# SQ counters of the decode attention kernel at a given suffix length (PMC="..." picks the counters, one pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY} --kernel-trace --output-format csv -d gpurun_out/pmc_ad -- python3 profiles/src/attn_decode_bench.py ${STEPS:-511} > gpurun_out/pmc_ad.log 2>&1
f=$(find gpurun_out/pmc_ad -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:30]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, c in agg.items():
    if 'attn' not in n: continue
    print(n, "launches", len(next(iter(c.values()))), " ".join(f"{k}={sum(v)/len(v):.0f}" for k, v in sorted(c.items())))
PY
rm -rf gpurun_out/pmc_ad
