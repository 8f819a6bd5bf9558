#!/bin/bash
# round 4, run 10: two-rank SFT tests, then the SFT step at 7B (128 frames of the reference geometry, LoRA recipe and full fine-tune)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sft_gpu.py -m gpu -x -q -k two_rank > gpurun_out/r04_sft_dp.log 2>&1
echo "dp rc=$?"; tail -5 gpurun_out/r04_sft_dp.log
timeout -k 10 900 python profiles/src/sft_bench.py 128 512 4 3 lora,full > gpurun_out/r04_sft_bench.log 2>&1
echo "bench rc=$?"; tail -6 gpurun_out/r04_sft_bench.log
