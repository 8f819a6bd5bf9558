#!/bin/bash
# round 4, run 12: the GRPO shell entry at 7B with LoRA adapters (rank 64 on every LM linear layer, LM frozen): 3 steps, 4 groups x 8 x 256
mkdir -p gpurun_out
export VR2_ENTRY_FLAGS="--freeze_llm True --lora_enable True --lora_rank 64 --lora_alpha 64 --lora_dropout 0.05 --lora_namespan_exclude ['lm_head','embed_tokens'] --learning_rate 1e-4"
timeout -k 10 1000 python profiles/src/full_size_entry.py /tmp/w7 3 4 256 480 > gpurun_out/r04_entry7b_lora.log 2>&1
echo "lora rc=$?"
grep -a "^{'step'\|entry7b\|LoRA" gpurun_out/r04_entry7b_lora.log | cut -c1-900
tail -5 gpurun_out/r04_entry7b_lora.log | cut -c1-600
