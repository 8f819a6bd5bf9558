#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python profiles/src/pmc_prefill.py > gpurun_out/pmc_mfma.log 2>&1
f=$(find gpurun_out/pmc_mfma -name "*counter_collection.csv" | head -1)
k=$(find gpurun_out/pmc_mfma -name "*kernel_trace.csv" | head -1)
python - "$f" "$k" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[n] += 1
# kernel durations
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    dur[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:12]:
    g = c.get('GRBM_GUI_ACTIVE', 0); m = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    # MfmaUtil = MFMA_BUSY / (GUI_ACTIVE(max over XCD; reported as sum over 8 XCDs -> /8) * SIMD_NUM=1024) * 100
    util = 100 * m / (g / 8 * 1024) if g else 0
    print(f"{n:42s} launches {cnt[n]:5d} dur_us {dur[n]:10.1f} MFMA_BUSY {m:14.0f} GUI_ACTIVE {g:12.0f} MfmaUtil% {util:6.1f}")
PY
rm -rf gpurun_out/pmc_mfma
