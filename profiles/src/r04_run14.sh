#!/bin/bash
# round 4, run 14: more sequences per decode weight pass (the 12- / 16-row-block ring GEMV): P = 24 (192 rows) and P = 32 (256 rows)
# against P = 16, 3 timed steps each, no CPU baseline / extras
mkdir -p gpurun_out
for P in 32 24 16; do
  timeout -k 10 420 python bench.py --gpus 1 --steps 3 --warmup 1 --prompts-per-gpu $P --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_bench_P$P.json 2> gpurun_out/r04_bench_P$P.err
  echo "P=$P rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_bench_P$P.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","hbm_free_GB","rollout_samples_per_s")}, d["roofline_decode"]["frac"], d["roofline_decode"]["avg_launch_us"], d.get("phase_ms"))
except Exception as e:
    print("no json", e); print(open("gpurun_out/r04_bench_P$P.err").read()[-1500:])
PY
done
