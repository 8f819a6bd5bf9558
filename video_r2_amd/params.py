"""Flag surface of the GRPO entry point: every flag of /root/reference/train/scripts/train_grpo.sh:41-80 is accepted with
the reference's spelling and value format ("True"/"False" strings, comma lists).  Flags that only configure the parts this
engine replaces (--deepspeed, --disable_flash_attn2, --gradient_checkpointing, --tf32, --lazy_preprocess ...) are accepted
and ignored; unknown flags do not crash (SURVEY.md 8(b) CLI seam).  Defaults follow train/src/params.py:120-219 and the
TRL GRPOConfig values listed in SURVEY.md Appendix A; ``loss_type`` is explicit because its TRL default is
version-dependent."""
from __future__ import annotations

import argparse
from dataclasses import dataclass, field, fields
from typing import Optional


def _bool(v):
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() in ("1", "true", "yes", "y", "t")


@dataclass
class GRPOArguments:
    # model / data
    model_id: str = "Qwen/Qwen2-VL-7B-Instruct"
    data_path: Optional[str] = None
    image_folder: Optional[str] = None
    video_subtitles_folder: Optional[str] = None
    output_dir: str = "output"
    image_min_pixels: int = 3136
    image_max_pixels: int = 12845056
    video_min_pixels: int = 100352
    video_max_pixels: int = 602112
    image_resized_width: Optional[int] = None
    image_resized_height: Optional[int] = None
    video_resized_width: Optional[int] = None
    video_resized_height: Optional[int] = None
    fps: float = 1.0                  # DataArguments default (params.py:218); train_grpo.sh passes 2.0
    # freezing (train_grpo.sh:47-49)
    freeze_vision_tower: bool = False
    freeze_llm: bool = False
    freeze_merger: bool = False
    bf16: bool = True
    fp16: bool = False
    # LoRA (train_grpo.py:76-93,137-149 of the reference; params.py:147-158)
    lora_enable: bool = False
    vision_lora: bool = False
    use_dora: bool = False
    lora_rank: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_weight_path: str = ""
    lora_bias: str = "none"
    lora_namespan_exclude: Optional[str] = None
    num_lora_modules: int = -1
    bits: int = 16
    # GRPO
    num_train_epochs: float = 1.0
    max_steps: int = -1
    num_generations: int = 8
    num_iterations: int = 1
    beta: float = 0.04
    epsilon: float = 0.2
    epsilon_high: Optional[float] = None
    loss_type: str = "bnpo"
    scale_rewards: bool = True
    mask_truncated_completions: bool = False
    sync_ref_model: bool = False      # TRL GRPOConfig: every ref_model_sync_steps steps ref <- alpha * policy + (1 - alpha) * ref
    ref_model_mixup_alpha: float = 0.6   # (SyncRefModelCallback, reached at grpo_trainer.py:587-588)
    ref_model_sync_steps: int = 512
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    max_completion_length: int = 256
    max_prompt_length: int = 512
    temperature: float = 0.9
    top_p: float = 1.0
    top_k: int = 50
    min_p: Optional[float] = None
    repetition_penalty: float = 1.0
    seed: int = 42
    shuffle_dataset: bool = True
    # rewards
    reward_llm_judge: str = "Qwen/Qwen3-0.6B"
    reward_func_names: str = "accuracy_reward,format_reward"
    reward_weights: str = "1.0,1.0"
    positive_reinforcement: str = ""
    length_reinforcement_buffer: int = 100
    buffer_seconds: int = 2
    similarity_threshold: float = 0.75
    reward_module: str = ""           # module holding the *_reward functions; "" = src.train.reward_funcs if importable, else the built-in one
    sentence_model_name: str = "all-MiniLM-L6-v2"   # embedding model handed to the TAR rewards (grpo_trainer.py:387)
    log_completions: bool = False
    # optimiser
    learning_rate: float = 1e-6
    weight_decay: float = 0.01
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "cosine"
    warmup_ratio: float = 0.0
    # bookkeeping
    logging_steps: int = 10
    save_strategy: str = "steps"
    save_steps: int = 500
    save_total_limit: int = 2
    report_to: str = "none"
    dataloader_num_workers: int = 0
    remove_unused_columns: bool = False
    resume_from_checkpoint: str = ""  # "" = resume from the newest output_dir/checkpoint-* if there is one (train_grpo.py:184-187)
    # engine options without a counterpart in the reference
    loss_norm_unit: str = "per_device_batch"   # token-mean losses (bnpo, mapo_bnpo) + kl / clip metrics are normalised over blocks of
                                      # per_device_train_batch_size completions and averaged, as the reference's ranks do;
                                      # "rank": over all completions a rank holds (round-1 behaviour of this engine)
    decode_weights: str = "bf16"     # "fp8": rollout decode on per-channel e4m3 weight copies and (above 16 rows) per-row e4m3 activations
                                      # on the fp8 MFMA (BASELINE config 5); "fp8w": weight-only fp8
    pack_completions: bool = True    # score / train on each completion up to its first EOS only (rows behind it are masked out of the
                                      # loss anyway): sum of lengths rows per group instead of G x max length
    profile_phases: bool = True      # profiling/Time taken: QwenGRPOTrainer.<phase> metrics (HIP events on the engine stream) + roctx ranges
    async_rewards: bool = True        # reward functions run on a background thread behind the reference-model scoring
    grad_exchange_dtype: str = "fp32"  # wire format of the data-parallel gradient all-reduce: fp32 (exact mean of the ranks' fp32 gradients,
                                      # 30.5 GB per step at 7B) or bf16 (15.2 GB: each summand rounded once, as the reference's bf16 ZeRO-3)
    reward_workers: int = 4           # each reward function scores this many slices of the step's completions concurrently (1 = one call)
    ignored: dict = field(default_factory=dict)


IGNORED_FLAGS = ("deepspeed", "disable_flash_attn2", "gradient_checkpointing", "tf32", "lazy_preprocess", "cache_dir", "optim",
                 "use_liger", "local_rank", "double_quant", "quant_type", "max_seq_length")


@dataclass
class SFTArguments:
    """train/src/params.py:14-59 (TrainingArguments over HF's) + DataArguments + ModelArguments, as train_sft.sh:27-66 sets them"""
    model_id: str = "Qwen/Qwen2-VL-7B-Instruct"
    data_path: Optional[str] = None
    image_folder: Optional[str] = None
    video_subtitles_folder: Optional[str] = None
    output_dir: str = "output"
    image_min_pixels: int = 3136
    image_max_pixels: int = 12845056
    video_min_pixels: int = 100352
    video_max_pixels: int = 602112
    image_resized_width: Optional[int] = None
    image_resized_height: Optional[int] = None
    video_resized_width: Optional[int] = None
    video_resized_height: Optional[int] = None
    fps: float = 1.0
    freeze_vision_tower: bool = False
    freeze_llm: bool = False
    freeze_merger: bool = False
    bf16: bool = True
    fp16: bool = False
    # LoRA (train_sft.sh:30-37)
    lora_enable: bool = False
    vision_lora: bool = False
    use_dora: bool = False
    lora_rank: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_weight_path: str = ""
    lora_bias: str = "none"
    lora_namespan_exclude: Optional[str] = None
    num_lora_modules: int = -1
    bits: int = 16
    # schedule / optimiser (HF TrainingArguments names and defaults; train_sft.sh:50-66)
    num_train_epochs: float = 1.0
    max_steps: int = -1
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    learning_rate: float = 5e-5
    vision_lr: Optional[float] = None
    merger_lr: Optional[float] = None
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "cosine"
    warmup_ratio: float = 0.0
    warmup_steps: int = 0
    seed: int = 42
    logging_steps: int = 10
    save_strategy: str = "steps"
    save_steps: int = 500
    save_total_limit: int = 2
    report_to: str = "none"
    dataloader_num_workers: int = 0
    remove_unused_columns: bool = False
    resume_from_checkpoint: str = ""
    grad_exchange_dtype: str = "fp32"
    ignored: dict = field(default_factory=dict)


def parse_sft_args(argv=None) -> SFTArguments:
    a = _parse(SFTArguments, argv, IGNORED_FLAGS)
    _check_lora(a)
    if a.grad_exchange_dtype not in ("fp32", "bf16"):
        raise ValueError(f"--grad_exchange_dtype must be fp32 or bf16, got {a.grad_exchange_dtype}")
    if a.lr_scheduler_type not in ("cosine", "constant"):
        raise NotImplementedError(f"--lr_scheduler_type {a.lr_scheduler_type}: cosine (the shipped recipe) and constant are built")
    return a


def _parse(cls, argv, ignored_flags):
    p = argparse.ArgumentParser(allow_abbrev=False)
    for f in fields(cls):
        if f.name == "ignored":
            continue
        typ = f.type if isinstance(f.type, type) else None
        default = f.default
        if isinstance(default, bool):
            p.add_argument(f"--{f.name}", type=_bool, default=default, nargs="?", const=True)
        elif isinstance(default, int):
            p.add_argument(f"--{f.name}", type=int, default=default)
        elif isinstance(default, float):
            p.add_argument(f"--{f.name}", type=float, default=default)
        elif default is None and "int" in str(f.type):
            p.add_argument(f"--{f.name}", type=lambda v: None if v in ("None", "") else int(v), default=None)
        elif default is None and "float" in str(f.type):
            p.add_argument(f"--{f.name}", type=lambda v: None if v in ("None", "") else float(v), default=None)
        else:
            p.add_argument(f"--{f.name}", type=str, default=default)
    for name in ignored_flags:
        p.add_argument(f"--{name}", type=str, default=None, nargs="?", const="True")
    ns, unknown = p.parse_known_args(argv)
    d = vars(ns)
    ignored = {k: d.pop(k) for k in ignored_flags if k in d}
    # unknown flags: keep "--name value" pairs for the record, never crash
    i = 0
    while i < len(unknown):
        tok = unknown[i]
        if tok.startswith("--"):
            val = unknown[i + 1] if i + 1 < len(unknown) and not unknown[i + 1].startswith("--") else "True"
            ignored[tok[2:]] = val
            i += 2 if val is not unknown[i] and i + 1 < len(unknown) and not unknown[i + 1].startswith("--") else 1
        else:
            i += 1
    a = cls(**d)
    a.ignored = {k: v for k, v in ignored.items() if v is not None}
    return a


def _check_lora(a):
    """the reference's own checks (train_grpo.py:76-93, train_sft.py:88-108) + what this engine's LoRA covers"""
    if a.lora_enable and not a.freeze_llm:
        raise ValueError("If `lora_enable` is True, `freeze_llm` must also be True.")
    if a.vision_lora and not a.lora_enable:
        raise ValueError("`vision_lora` needs `lora_enable`.")
    if a.vision_lora and not a.freeze_vision_tower:
        raise ValueError("If `vision_lora` is True, `freeze_vision_tower` must also be True.")
    if a.lora_enable:
        if a.vision_lora or a.use_dora or a.lora_bias != "none" or a.bits != 16:
            raise NotImplementedError("LoRA on this engine: adapters on the LM's linear layers, bias none, 16-bit base weights "
                                      "(--vision_lora / --use_dora / --lora_bias / --bits 4|8 are not built)")
        import ast
        ex = ast.literal_eval(a.lora_namespan_exclude) if a.lora_namespan_exclude else []
        if not {"lm_head", "embed_tokens"} <= set(ex):
            raise NotImplementedError("LoRA on this engine leaves lm_head and embed_tokens without adapters: pass "
                                      "--lora_namespan_exclude \"['lm_head', 'embed_tokens']\" (the shipped recipe's value)")


def parse_args(argv=None) -> GRPOArguments:
    a = _parse(GRPOArguments, argv, IGNORED_FLAGS)
    if a.epsilon_high in (None, "None", ""):
        a.epsilon_high = None
    elif not isinstance(a.epsilon_high, float):
        a.epsilon_high = float(a.epsilon_high)
    _check_lora(a)
    if a.grad_exchange_dtype not in ("fp32", "bf16"):
        raise ValueError(f"--grad_exchange_dtype must be fp32 or bf16, got {a.grad_exchange_dtype}")
    if a.loss_norm_unit not in ("per_device_batch", "rank"):
        raise ValueError(f"--loss_norm_unit must be per_device_batch or rank, got {a.loss_norm_unit}")
    if a.num_generations < 2:
        raise ValueError("GRPO requires at least 2 generations per prompt to calculate the advantages.")
    return a
