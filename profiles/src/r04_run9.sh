#!/bin/bash
# round 4, run 9: the shell entry at 7B with the vision tower and the merger trainable (2 steps, 4 groups x 8 x 256 tokens), then the
# same with both frozen for the per-phase comparison
mkdir -p gpurun_out
export VR2_ENTRY_FLAGS="--freeze_vision_tower False --freeze_merger False"
timeout -k 10 900 python profiles/src/full_size_entry.py /tmp/w7 2 4 256 480 > gpurun_out/r04_entry7b_vision.log 2>&1
echo "vision rc=$?"
grep -a "^{'step'\|entry7b\|trainable" gpurun_out/r04_entry7b_vision.log | cut -c1-1500
export VR2_ENTRY_FLAGS=""
rm -rf /tmp/w7/out
timeout -k 10 600 python profiles/src/full_size_entry.py /tmp/w7 2 4 256 480 > gpurun_out/r04_entry7b_frozen.log 2>&1
echo "frozen rc=$?"
grep -a "^{'step'\|entry7b" gpurun_out/r04_entry7b_frozen.log | cut -c1-1500
