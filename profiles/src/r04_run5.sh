#!/bin/bash
# round 4, GPU call 7: PMC on the final tree, the whole GPU suite, smoke, the default bench as the driver runs it
mkdir -p gpurun_out
bash profiles/src/r04_pmc_only.sh > gpurun_out/r04_pmc_only.log 2>&1; tail -25 gpurun_out/r04_pmc_only.log
python -m pytest tests -m gpu -q > gpurun_out/r04_t6.log 2>&1; echo rc=$? >> gpurun_out/r04_t6.log; tail -4 gpurun_out/r04_t6.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04_smoke.log
python bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo "bench rc=$?"; tail -2 gpurun_out/r04_bench_final.err; cut -c1-120 gpurun_out/r04_bench_final.json
