#!/bin/bash
# round 4, run 18 (final tree, default P = 22): the GPU suite, then the default bench as the driver runs it (20 timed steps after 5 warm-up steps)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_full_gpu_suite5.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r04_full_gpu_suite5.log
date +%s > gpurun_out/r04_t0.txt
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_final_p22_20steps.json 2> gpurun_out/r04_bench_final_p22_20steps.err; echo "bench rc=$? wall $(( $(date +%s) - $(cat gpurun_out/r04_t0.txt) )) s"
tail -2 gpurun_out/r04_bench_final_p22_20steps.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_final_p22_20steps.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","hbm_free_GB","recipe_shape_samples_per_s","rollout_samples_per_s")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_decode"]["frac"], d["roofline_decode"]["traffic"], d["roofline_wgrad"]["frac"], d["cpu_baseline"]["value"], d.get("eos_mix",{}).get("samples_per_s"), d["phase_ms"])
PY
