#!/bin/bash
# round 4, GPU call 4: zero-operand GEMM (power / clock ceiling), GPU test suite, default bench
mkdir -p gpurun_out
( cd profiles/src && VR2_BENCH_V7=1 VR2_BENCH_SHAPE="lm gate/up" timeout -k 10 120 ./gemm_bench && VR2_BENCH_ZERO=1 VR2_BENCH_V7=1 VR2_BENCH_SHAPE="lm gate/up" timeout -k 10 120 ./gemm_bench ) > gpurun_out/r04_gemm_zero.log 2>&1
grep "v5\|v7" gpurun_out/r04_gemm_zero.log
python -m pytest tests -m gpu -q -s > gpurun_out/r04_t3.log 2>&1; echo rc=$? >> gpurun_out/r04_t3.log
tail -8 gpurun_out/r04_t3.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err; echo "bench rc=$?"
tail -3 gpurun_out/r04_bench1.err; cut -c1-600 gpurun_out/r04_bench1.json
