#!/bin/bash
# GRPO stage on MI355X.  Entry point and flag names follow the reference recipe so existing launch tooling keeps working;
# underneath: one process per GPU (torch.distributed.run), replicated weights, prompt groups sharded over the ranks,
# one RCCL gradient all-reduce per optimiser step.  Every flag can be overridden by appending it ("$@" wins).
#
# Batch shape: the batch flags count completions per device as in the reference.  per_device_train_batch_size *
# gradient_accumulation_steps >= num_generations: a rank owns whole prompt groups (that many / num_generations of them).
# FEWER than num_generations (the recipe values below: 1 per device, 8 generations): one prompt group is spread over
# num_generations / per_device ranks, each generating its slice of the completions; rewards are gathered and the advantages
# taken over the whole group -- on 8 GPUs that is ONE prompt per optimiser step, the reference's batch.  For throughput raise
# per_device_train_batch_size to num_generations * P: P groups per GPU are decoded together (the decode streams the weights
# once for up to 256 sequences; a 1-row decode streams them for a single sequence).
#
# Rewards: the functions are looked up by name in --reward_module (default: your src/train/reward_funcs.py next to this
# entry point if it imports, else the engine's built-in accuracy / format rewards).  The temporal-grounding reward named
# below is the reference's own code (LLM judge + MiniLM): keep your reward_funcs.py in place, or drop the name.
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0                       # dmabuf IPC for RCCL on this driver

: "${NUM_DEVICES:=8}"
: "${MODEL_NAME:=path/to/the/Video-R2-SFT-Model}"          # HF Qwen2_5_VLForConditionalGeneration directory
: "${DATA_PATH:=path/to/video-r2-grpo-dataset.json}"
: "${DATA_FOLDER:=path/to/videos}"
: "${SUBTITLES_FOLDER:=path/to/subtitles}"
: "${OUTPUT_DIR:=path/to/output_folder}"

# host video front end (read by the dataset code, names as in the reference)
export FPS_MAX_FRAMES="${FPS_MAX_FRAMES:-32}"
export WRITE_TIMESTAMPS_ON_FRAMES="${WRITE_TIMESTAMPS_ON_FRAMES:-True}"
export WRITE_SUBTITLES_ON_FRAMES="${WRITE_SUBTITLES_ON_FRAMES:-True}"
# OpenAI-compatible judge endpoint for the LLM-scored rewards
export SERVED_MODEL="${SERVED_MODEL:-Qwen/Qwen3-Next-80B-A3B-Instruct}"
export OPENAI_API_BASE="${OPENAI_API_BASE:-http://${LLM_SERVER_ADDRESS:-localhost}:${SERVER_PORT:-8011}/v1}"

model_flags=(
    --model_id "$MODEL_NAME"
    --freeze_llm False                 # the LM is what GRPO trains ...
    --freeze_vision_tower True         # ... the ViT and the merger stay frozen
    --freeze_merger True
    --bf16 True --fp16 False --tf32 False
    --disable_flash_attn2 False        # accepted, ignored: attention is the engine's own
    --gradient_checkpointing True      # accepted, ignored: activations are kept (288 GB HBM)
    --deepspeed scripts/zero3.json     # accepted, ignored: weights are replicated
)
data_flags=(
    --data_path "$DATA_PATH"
    --image_folder "$DATA_FOLDER"
    --video_subtitles_folder "$SUBTITLES_FOLDER"
    --fps 2.0
    --video_max_pixels $((360 * 420))
    --lazy_preprocess True
    --dataloader_num_workers 4
    --remove_unused_columns False
)
grpo_flags=(
    --num_generations 8
    --max_prompt_length 32768
    --max_completion_length 1024
    --beta 0.04
    --reward_func_names "accuracy_reward,format_reward,temporal_grounding_sentence_embedding_consistency_reward"
    --reward_weights 1.0,1.0,1.0
    --reward_llm_judge ''
    --similarity_threshold 0.75
    --buffer_seconds 2
)
optim_flags=(
    --num_train_epochs 1
    --per_device_train_batch_size 1
    --gradient_accumulation_steps 1
    --learning_rate 1e-6
    --weight_decay 0.01
    --lr_scheduler_type cosine
)
log_flags=(
    --output_dir "$OUTPUT_DIR"
    --logging_steps 10
    --report_to tensorboard
    --save_strategy steps --save_steps 500 --save_total_limit 2
)

exec python -m torch.distributed.run --nnodes=1 --nproc-per-node="$NUM_DEVICES" --master-addr 127.0.0.1 \
    src/train/train_grpo.py "${model_flags[@]}" "${data_flags[@]}" "${grpo_flags[@]}" "${optim_flags[@]}" "${log_flags[@]}" "$@"
