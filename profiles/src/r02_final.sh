#!/bin/bash
# round 2, final state: (1) kernel stats of the default bench, (2) fabric traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
# MFMA busy cycles of the training GEMMs at T = 9280 rows, (3) the config-5 bench line.  Summaries -> gpurun_out/r02f_*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash profiles/src/prof_step.sh r02f > gpurun_out/r02f_step.log 2>&1
echo "stats done" > gpurun_out/r02f_progress.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_$tag -- python3 profiles/src/pmc_train_gemms.py > gpurun_out/pmcf_$tag.log 2>&1
  f=$(find gpurun_out/pmcf_$tag -name "*counter_collection.csv" | head -1)
  k=$(find gpurun_out/pmcf_$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$k" >> gpurun_out/r02f_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:44] + " grid " + r.get('Grid_Size', '?')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:44] + " grid " + r.get('Grid_Size', '?')
    dur[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, c in agg.items():
    if 'gemm' not in n: continue
    parts = [f"{k} mean {sum(v)/len(v):.1f}" for k, v in c.items()]
    d = dur.get(n, [0]); 
    print(n, "launches", len(next(iter(c.values()))), "mean_us %.1f" % (sum(d) / len(d)), *parts)
PY
  rm -rf gpurun_out/pmcf_$tag
  echo "$tag done" >> gpurun_out/r02f_progress.txt
done
python3 bench.py --decode-weights fp8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02f_config5.json 2> gpurun_out/r02f_config5.err
echo "config5 done" >> gpurun_out/r02f_progress.txt
cat gpurun_out/r02f_pmc.txt
