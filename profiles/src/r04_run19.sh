#!/bin/bash
# round 4, run 19: the fp8 x fp8 ring GEMV at 129-256 rows: kernel tests, the engine's fp8 decode tests, then config 5 at P = 16 / 22 / 32
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "fp8 or gemv or config5" > gpurun_out/r04_t19.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_t19.log
for P in 22 32 16; do
  timeout -k 10 420 python bench.py --gpus 1 --steps 3 --warmup 1 --decode-weights fp8 --prompts-per-gpu $P --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_bench_cfg5_P$P.json 2> gpurun_out/r04_bench_cfg5_P$P.err
  echo "cfg5 P=$P rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_bench_cfg5_P$P.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","hbm_free_GB")}, d["roofline_decode"]["kernel"][:60], d["roofline_decode"]["frac"], d["roofline_decode"]["avg_launch_us"])
except Exception as e:
    print("no json", e); print(open("gpurun_out/r04_bench_cfg5_P$P.err").read()[-1500:])
PY
done
