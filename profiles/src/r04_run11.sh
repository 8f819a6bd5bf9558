#!/bin/bash
# round 4, run 11: rocprofv3 kernel stats of the 7B SFT step (128 frames of the reference geometry, 2 rows per step, 1 timed step after the
# warm-up step), LoRA recipe and full fine-tune in processes of their own
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mode in lora full; do
tag=r04_sft_$mode
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 profiles/src/sft_bench.py 128 512 2 1 $mode > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:64]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}.json
head -30 gpurun_out/${tag}_kernel_stats.txt
done
