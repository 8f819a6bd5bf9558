"""Stage-1 SFT step at full size on one MI355X: Qwen2.5-VL-7B (random init), rows of ONE video at the reference's SFT geometry
(FPS_MAX_FRAMES frames of 280 x 504 -> frames / 2 * 180 video tokens) + a 64-token question + an ANSWER-token answer, the shipped
LoRA recipe (rank 64 on every LM linear layer, everything else frozen) and the full fine-tune of the LM beside it.
Usage: python profiles/src/sft_bench.py [frames=128] [answer_tokens=512] [rows_per_step=4] [steps=3] [modes=lora,full]
Prints one JSON line per mode: tokens/s (all sequence positions of the step's rows / wall second), ms per row, phase split."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import model_cfg  # noqa: E402
from video_r2_amd import params, weights as W  # noqa: E402
from video_r2_amd.model import VideoR2Model  # noqa: E402
from video_r2_amd.sft import QwenSFTTrainer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
answer = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rows_per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
modes = (sys.argv[5] if len(sys.argv) > 5 else "lora,full").split(",")
cfg = model_cfg("7b")
dev = "cuda:0"
sd = W.random_state_dict(cfg, seed=0, device=dev)
g = torch.Generator().manual_seed(0)


def row(i):
    fr = torch.randint(0, 256, (frames, 3, 280, 504), generator=g, dtype=torch.uint8)
    tv = frames // 2 * (280 // 14) * (504 // 14) // 4
    q = torch.randint(1000, 150000, (64,), generator=g).tolist()
    a = torch.randint(1000, 150000, (answer,), generator=g).tolist() + [cfg.eos_token_id]
    ids = [cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id] + q
    return dict(input_ids=np.asarray(ids + a), labels=np.asarray([-100] * len(ids) + a), videos=[(fr.pin_memory(), None)], second_per_grid_ts=[2.0])


rows = [row(i) for i in range(rows_per_step)]
for mode in modes:
    m = VideoR2Model.from_state_dict(cfg, sd, device=dev)
    flags = ["--per_device_train_batch_size", "1", "--gradient_accumulation_steps", str(rows_per_step), "--learning_rate", "1e-5", "--weight_decay", "0.1",
             "--warmup_ratio", "0.03", "--max_steps", str(steps + 1), "--freeze_vision_tower", "True", "--freeze_merger", "True"]
    lora = None
    if mode == "lora":
        from video_r2_amd.lora import LoraTrainer
        flags += ["--freeze_llm", "True", "--lora_enable", "True", "--lora_rank", "64", "--lora_alpha", "64", "--lora_dropout", "0.05",
                  "--lora_namespan_exclude", "['lm_head', 'embed_tokens']"]
        lora = LoraTrainer(m.engine, m.text, rank=64, alpha=64, dropout=0.05, seed=42)
    args = params.parse_sft_args(flags)
    tr = QwenSFTTrainer(m, args, list(range(64)), lambda ex: ex, lora=lora)
    mbs = [[r] for r in rows]
    tr.step(mbs, mbs)                                   # warm-up: code objects, workspaces
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        out = tr.step(mbs, mbs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    toks = sum(len(r["input_ids"]) for r in rows)
    print(json.dumps({"mode": mode, "frames": frames, "tokens_per_row": toks // rows_per_step, "rows_per_step": rows_per_step, "s_per_step": round(dt, 3),
                      "tokens_per_s": round(toks / dt, 1), "ms_per_row": round(1e3 * dt / rows_per_step, 1), "loss": out["loss"],
                      "trainable_params": (lora.n_params() if lora else tr.updater.state.n_params()),
                      "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
    del tr, m, lora
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
