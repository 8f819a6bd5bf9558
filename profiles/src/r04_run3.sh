#!/bin/bash
# round 4, GPU call 5: new tests, config-1 bench with the CPU baseline in full, profiles + PMC
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_eval_adapter_gpu.py tests/test_7b_width_oracle_gpu.py tests/test_model_gpu.py -m gpu -q -s > gpurun_out/r04_t4.log 2>&1; echo rc=$? >> gpurun_out/r04_t4.log
tail -6 gpurun_out/r04_t4.log
python bench.py --model 3b --frames 4 --G 2 --max-new 32 --prompts-per-gpu 1 --cpu-baseline full --steps 5 --warmup 2 --eos-mix off > gpurun_out/r04_bench_config1_full_cpu.json 2> gpurun_out/r04_bench_config1_full_cpu.err; echo "config1 rc=$?"
tail -2 gpurun_out/r04_bench_config1_full_cpu.err; cut -c1-300 gpurun_out/r04_bench_config1_full_cpu.json
bash profiles/src/r04_profiles.sh > gpurun_out/r04_profiles.log 2>&1; echo "profiles rc=$?"
tail -60 gpurun_out/r04_profiles.log
