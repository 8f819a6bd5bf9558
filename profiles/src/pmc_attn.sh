#!/bin/bash
# LDS bank-conflict cycles of the head_dim-128 attention kernels (forward, dQ, dK/dV) at the training row space
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_attn -- python3 profiles/src/attn_train_bench.py > gpurun_out/pmc_attn.log 2>&1
f=$(find gpurun_out/pmc_attn -name "*counter_collection.csv" | head -1)
python3 - "$f" > gpurun_out/pmc_attn.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, c in agg.items():
    if 'attn' not in n and 'gemm_tn' not in n: continue
    bc = sum(c.get('SQ_LDS_BANK_CONFLICT', [0])) ; ia = sum(c.get('SQ_LDS_IDX_ACTIVE', [0]))
    print(f"{n:42s} launches {len(c.get('SQ_LDS_IDX_ACTIVE', []))} LDS_BANK_CONFLICT {bc:.0f} LDS_IDX_ACTIVE {ia:.0f} conflict share {100*bc/max(ia,1):.2f} %")
PY
rm -rf gpurun_out/pmc_attn
cat gpurun_out/pmc_attn.txt
