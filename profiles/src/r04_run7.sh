#!/bin/bash
# round 4, GPU call 9: kernel stats of the recipe-shape decode (8 rows); the entry test; the bench as the driver runs it (25 timed
# steps).  (A --pmc GRBM_GUI_ACTIVE pass over the whole bench, for the clock per kernel family inside the step, made rocprofv3 itself
# segfault and was dropped: the effective clock of the long GEMM launches comes from the PMC passes over pmc_train_gemms.py instead.)
mkdir -p gpurun_out
bash profiles/src/prof_dec64.sh 1 256 > gpurun_out/r04_dec8rows.log 2>&1; head -22 gpurun_out/r04_dec8rows.log
python -m pytest tests/test_entry_gpu.py -m gpu -q > gpurun_out/r04_t8.log 2>&1; tail -3 gpurun_out/r04_t8.log
python bench.py --gpus 1 --steps 25 --warmup 2 > gpurun_out/r04_bench_25steps.json 2> gpurun_out/r04_bench_25steps.err; echo "bench rc=$?"; cut -c1-120 gpurun_out/r04_bench_25steps.json
