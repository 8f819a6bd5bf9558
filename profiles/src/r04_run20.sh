#!/bin/bash
# round 4, run 20 (final tree): BASELINE configs 4 and 2 (3 timed steps each)
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --frames 64 --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_bench_config4_final.json 2> gpurun_out/r04_bench_config4_final.err; echo "cfg4 rc=$?"
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --frames 16 --G 4 --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_bench_config2_final.json 2> gpurun_out/r04_bench_config2_final.err; echo "cfg2 rc=$?"
python - <<PY
import json
for n in ("config4","config2"):
    try:
        d=json.loads(open(f"gpurun_out/r04_bench_{n}_final.json").read().strip().splitlines()[-1])
        print(n, {k:d.get(k) for k in ("value","ms_per_step","decode_ms_per_token","hbm_peak_GB","hbm_free_GB")}, d["config"]["workload"])
    except Exception as e:
        print(n, "no json", e, open(f"gpurun_out/r04_bench_{n}_final.err").read()[-800:])
PY
