#!/bin/bash
# round 3: (1) kernel stats of the default bench (3 steps traced), (2) kernel stats of the config-5 bench (fp8 decode + fp8 no-grad
# GEMMs), (3) fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA busy cycles of vr2_gemm_a8w8 next to vr2_gemm_bf16.
# Summaries -> gpurun_out/r03_*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
stats() {   # $1 = tag, rest = bench flags
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-recipe-shape --eos-mix off "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" > gpurun_out/${tag}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:48]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:64]
    print(f"| {n} | {int(r['Calls'])} | {int(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
PY
  rm -rf gpurun_out/prof_$tag
  echo "$tag stats done" >> gpurun_out/r03_progress.txt
}
stats r03_default
stats r03_config5 --decode-weights fp8
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc3_$tag -- python3 profiles/src/pmc_gemm_fp8.py > gpurun_out/pmc3_$tag.log 2>&1
  f=$(find gpurun_out/pmc3_$tag -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmc3_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$k" >> gpurun_out/r03_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52] + " grid " + r.get('Grid_Size', '?')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:52] + " grid " + r.get('Grid_Size', '?')
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, c in agg.items():
    if 'gemm' not in n: continue
    parts = [f"{k} mean {sum(v[1:])/max(len(v)-1,1):.1f}" for k, v in c.items()]
    d = dur.get(n, [0, 0])
    print(n, "launches", len(next(iter(c.values()))), "mean_us(after first) %.1f" % (sum(d[1:]) / max(len(d) - 1, 1)), *parts)
PY
  rm -rf gpurun_out/pmc3_$tag
  echo "$tag done" >> gpurun_out/r03_progress.txt
done
cat gpurun_out/r03_pmc.txt
head -32 gpurun_out/r03_default_kernel_stats.txt
head -32 gpurun_out/r03_config5_kernel_stats.txt
