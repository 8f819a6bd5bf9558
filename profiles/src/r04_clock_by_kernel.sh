#!/bin/bash
# round 4: the clock the chip holds under each kernel family INSIDE the step: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8
# / kernel duration, one --pmc pass with --kernel-trace over a short bench (1 warm-up + 1 timed + 1 instrumented step, 256 new tokens so
# that the pass stays short); reads high on dispatches below ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back), so only the long kernels count
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_clk -- python3 bench.py --steps 1 --warmup 1 --max-new 256 --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_clk_bench.json 2> gpurun_out/r04_clk_bench.err
f=$(find gpurun_out/pmc_clk -name "*counter_collection.csv" | head -1)
k=$(find gpurun_out/pmc_clk -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$k" > gpurun_out/r04_clock_by_kernel.txt <<'PY'
import csv, sys, collections
gui = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        gui[r['Dispatch_Id']] = (r['Kernel_Name'], float(r['Counter_Value']))
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d, (name, g) in gui.items():
    if d not in dur: continue
    n = name.replace('(anonymous namespace)::', '').split('(')[0][:56]
    a = agg[n]; a[0] += g / 8.0; a[1] += dur[d]; a[2] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]
print("| kernel | launches | avg us | effective clock GHz (GUI_ACTIVE / 8 / duration) |")
print("|---|---|---|---|")
for n, (cyc, ns, c) in rows:
    print(f"| {n} | {c} | {ns / c / 1e3:.1f} | {cyc / ns:.2f} |")
PY
rm -rf gpurun_out/pmc_clk
cat gpurun_out/r04_clock_by_kernel.txt
