#!/bin/bash
# round 4, GPU call 8: the shell entry point at 7B (2 groups x 256 tokens: kl must be EXACTLY 0 at step 1 now; 16 groups x 1024 tokens:
# timing), decode attention chunk lengths re-swept with two suffix tiles in flight, the new config-5 ratio test
mkdir -p gpurun_out
W=/tmp/vr2_entry
python profiles/src/full_size_entry.py $W 2 2 256 480 > gpurun_out/r04_entry7b_2x256.log 2>&1; echo "entry 2x256 rc=$?"; grep -n "'kl'\|\"kl\"\|kl" gpurun_out/r04_entry7b_2x256.log | head -6
python profiles/src/full_size_entry.py $W 2 16 1024 480 > gpurun_out/r04_entry7b_16x1024.log 2>&1; echo "entry 16x1024 rc=$?"; tail -3 gpurun_out/r04_entry7b_16x1024.log
rm -rf $W
VR2_DEPTHS="2,1" VR2_DECODE_CHUNK="256;384;512;768" VR2_PREFIX_CHUNK="192;256;320;384" timeout -k 10 500 python profiles/src/attn_decode_bench.py > gpurun_out/r04_attn_chunks_depth2.log 2>&1
grep "==" gpurun_out/r04_attn_chunks_depth2.log
python -m pytest tests/test_model_gpu.py -m gpu -q -k "config5 or two_rank" > gpurun_out/r04_t7.log 2>&1; tail -3 gpurun_out/r04_t7.log
