#!/bin/bash
# round 4, GPU call 3: standalone GEMM v7 vs the shipped structure; decode attention tiles-in-flight sweep
mkdir -p gpurun_out
timeout -k 10 300 profiles/src/gemm_bench_run.sh > gpurun_out/r04_gemm_v7.log 2>&1; echo "gemm rc=$?" >> gpurun_out/r04_gemm_v7.log
tail -40 gpurun_out/r04_gemm_v7.log
VR2_DEPTHS="1,1;2,1;3,1;4,1;1,2;2,2;3,2;4,3" timeout -k 10 400 python profiles/src/attn_decode_bench.py > gpurun_out/r04_attn_depth.log 2>&1; echo "attn rc=$?" >> gpurun_out/r04_attn_depth.log
grep "==" gpurun_out/r04_attn_depth.log
