#!/bin/bash
# round 4, GPU call 6: kernel tests after the reduce change, eval throughput at the reference's 128-frame recipe, the other BASELINE configs
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_fullsize_7b_gpu.py -m gpu -q > gpurun_out/r04_t5.log 2>&1; echo rc=$? >> gpurun_out/r04_t5.log
tail -4 gpurun_out/r04_t5.log
python profiles/src/eval_128f.py --batch 8 --requests 8 --new 256 > gpurun_out/r04_eval_128f.json 2> gpurun_out/r04_eval_128f.err; echo "eval rc=$?"; tail -2 gpurun_out/r04_eval_128f.err; cat gpurun_out/r04_eval_128f.json
python profiles/src/eval_128f.py --batch 1 --requests 2 --new 256 > gpurun_out/r04_eval_128f_b1.json 2> gpurun_out/r04_eval_128f_b1.err; echo "eval b1 rc=$?"; cat gpurun_out/r04_eval_128f_b1.json
python bench.py --steps 3 --warmup 1 --decode-weights fp8 --no-cpu-baseline > gpurun_out/r04_bench_config5.json 2> gpurun_out/r04_bench_config5.err; echo "cfg5 rc=$?"; cut -c1-200 gpurun_out/r04_bench_config5.json
python bench.py --steps 3 --warmup 1 --frames 64 --no-cpu-baseline > gpurun_out/r04_bench_config4.json 2> gpurun_out/r04_bench_config4.err; echo "cfg4 rc=$?"; cut -c1-200 gpurun_out/r04_bench_config4.json
python bench.py --steps 3 --warmup 1 --frames 16 --G 4 --no-cpu-baseline > gpurun_out/r04_bench_config2.json 2> gpurun_out/r04_bench_config2.err; echo "cfg2 rc=$?"; cut -c1-200 gpurun_out/r04_bench_config2.json
