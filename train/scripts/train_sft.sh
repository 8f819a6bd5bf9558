#!/bin/bash
# Stage-1 (timestamp-aware) SFT on MI355X.  Entry point and flag names follow the reference recipe
# (scripts/train_sft.sh of the reference: LoRA r = 64 on the LM's linear layers, everything else frozen, global batch 32);
# underneath: one process per GPU (torch.distributed.run), replicated weights, rows sharded over the ranks, one RCCL gradient
# all-reduce per optimiser step.  Every flag can be overridden by appending it ("$@" wins).
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0                       # dmabuf IPC for RCCL on this driver

: "${GLOBAL_BATCH_SIZE:=32}"
: "${BATCH_PER_DEVICE:=1}"
: "${NUM_DEVICES:=8}"
GRAD_ACCUM_STEPS=$((GLOBAL_BATCH_SIZE / (BATCH_PER_DEVICE * NUM_DEVICES)))
: "${MODEL_NAME:=Qwen/Qwen2.5-VL-7B-Instruct}"            # HF Qwen2_5_VLForConditionalGeneration directory
: "${DATA_PATH:=path/to/video-r2-sft-dataset.json}"
: "${DATA_FOLDER:=path/to/videos}"
: "${SUBTITLES_FOLDER:=path/to/subtitles}"
: "${OUTPUT_DIR:=path/to/output_folder}"

export WRITE_TIMESTAMPS_ON_FRAMES="${WRITE_TIMESTAMPS_ON_FRAMES:-True}"
export WRITE_SUBTITLES_ON_FRAMES="${WRITE_SUBTITLES_ON_FRAMES:-True}"
export FPS_MAX_FRAMES="${FPS_MAX_FRAMES:-128}"            # up to 128 frames per video during SFT

model_flags=(
    --model_id "$MODEL_NAME"
    --lora_enable True --use_dora False
    --lora_namespan_exclude "['lm_head', 'embed_tokens']"
    --lora_rank 64 --lora_alpha 64 --lora_dropout 0.05 --num_lora_modules -1
    --freeze_vision_tower True --freeze_llm True --freeze_merger True
    --bf16 True --fp16 False --tf32 False
    --use_liger False                  # accepted, ignored: the kernels are the engine's own
    --disable_flash_attn2 False        # accepted, ignored
    --gradient_checkpointing True      # accepted, ignored: activations are kept (288 GB HBM)
    --deepspeed scripts/zero3.json     # accepted, ignored: weights are replicated
)
data_flags=(
    --data_path "$DATA_PATH"
    --image_folder "$DATA_FOLDER"
    --video_subtitles_folder "$SUBTITLES_FOLDER"
    --fps 1.0
    --video_max_pixels $((360 * 420))
    --lazy_preprocess True
    --dataloader_num_workers 4
    --remove_unused_columns False
)
train_flags=(
    --output_dir "$OUTPUT_DIR"
    --num_train_epochs 1
    --per_device_train_batch_size "$BATCH_PER_DEVICE"
    --gradient_accumulation_steps "$GRAD_ACCUM_STEPS"
    --learning_rate 1e-5 --merger_lr 1e-5 --vision_lr 2e-6
    --weight_decay 0.1 --warmup_ratio 0.03 --lr_scheduler_type cosine
    --logging_steps 1 --report_to tensorboard --save_strategy no
)
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node="$NUM_DEVICES" --master-addr "${MASTER_ADDR:-127.0.0.1}" \
    --master-port "${MASTER_PORT:-29517}" src/train/train_sft.py "${model_flags[@]}" "${data_flags[@]}" "${train_flags[@]}" "$@"
