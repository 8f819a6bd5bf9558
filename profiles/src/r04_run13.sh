#!/bin/bash
# round 4, run 13 (final tree): the new LoRA test, then the default bench as the driver runs it (20 timed steps after 5 warm-up steps)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_sft_gpu.py -m gpu -x -q -k num_lora > gpurun_out/r04_t13.log 2>&1; echo "test rc=$?"; tail -2 gpurun_out/r04_t13.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_final_tree_20steps.json 2> gpurun_out/r04_bench_final_tree_20steps.err; echo "bench rc=$?"
tail -2 gpurun_out/r04_bench_final_tree_20steps.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_final_tree_20steps.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","recipe_shape_samples_per_s") if k in d}, d["roofline"]["frac"], d["roofline_decode"]["frac"], d["roofline_wgrad"]["frac"], d["cpu_baseline"]["value"])
PY
