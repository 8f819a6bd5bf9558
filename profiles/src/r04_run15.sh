#!/bin/bash
# round 4, run 15: GEMV tests at every row-block count, then P = 20 and P = 22 (160 / 176 rows per decode weight pass)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemv" > gpurun_out/r04_t15.log 2>&1; echo "test rc=$?"; tail -2 gpurun_out/r04_t15.log
for P in 20 22; do
  timeout -k 10 420 python bench.py --gpus 1 --steps 3 --warmup 1 --prompts-per-gpu $P --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_bench_P$P.json 2> gpurun_out/r04_bench_P$P.err
  echo "P=$P rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_bench_P$P.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","hbm_free_GB")}, d["roofline_decode"]["frac"], d["roofline_decode"]["avg_launch_us"])
PY
done
