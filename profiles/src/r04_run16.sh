#!/bin/bash
# round 4, run 16: decode knobs at 192 rows (P = 24): suffix / prefix chunk lengths and split counts, rollout only (--no-update keeps the
# run short: decode ms/token is what is compared)
mkdir -p gpurun_out
run() {
  env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 2 --warmup 1 --prompts-per-gpu 24 --no-update --no-cpu-baseline --no-recipe-shape --eos-mix off > gpurun_out/r04_knob.json 2> gpurun_out/r04_knob.err
  python - "$*" <<PY
import json, sys
try:
    d=json.loads(open("gpurun_out/r04_knob.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "decode_ms_per_token", round(d["decode_ms_per_token"],3), "ms_per_step", round(d["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/r04_knob.err").read()[-600:])
PY
}
run VR2_NOP=1
run VR2_DECODE_CHUNK=512
run VR2_DECODE_CHUNK=256
run VR2_PREFIX_CHUNK=320
run VR2_PREFIX_CHUNK=384 VR2_DECODE_CHUNK=512
run VR2_SPLITS=4,12
run VR2_SPLITS=2,8
