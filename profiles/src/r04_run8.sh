#!/bin/bash
# round 4, GPU call 10: the split-group data-parallel test, the real-width gradient test, the DP tests
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py tests/test_7b_width_grads_gpu.py -m gpu -q -s > gpurun_out/r04_t9.log 2>&1; echo rc=$? >> gpurun_out/r04_t9.log
grep -n "7B widths\|passed\|failed\|FAILED\|Error" gpurun_out/r04_t9.log | head -20
