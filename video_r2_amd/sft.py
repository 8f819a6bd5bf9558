"""Stage-1 supervised fine-tuning on the engine: what `QwenSFTTrainer` (HF Trainer + the mixed-modality forward) does per step
in the reference (/root/reference/train/src/trainer/sft_trainer.py:31-110, train/monkey_patch_forward.py:497-648), on the kernels
of the GRPO update.

One optimiser step = `gradient_accumulation_steps` micro-batches of `per_device_train_batch_size` rows on every rank.  A micro-batch's
loss is the token mean of the cross entropy over its supervised positions (labels != -100 after the one-token shift,
monkey_patch_forward.py:619-631: `CrossEntropyLoss()` over the flattened shifted logits); HF Trainer divides each micro-batch loss
by the accumulation count and the data-parallel mean over ranks finishes the average -- so a row's log-prob gradient is
-1 / (supervised tokens of ITS micro-batch x accumulation steps), per rank, and the ranks' gradients are averaged.

A row runs as prompt = everything before its first supervised token, "completion" = the rest (one sequence): the unified row space,
the training forward with kept activations, the head's log-probs at the label tokens and the backward pass are train.Trainer's,
at temperature 1.  Vision tokens come from Engine.encode_videos, or -- when the tower / merger train -- from the patch rows through
train_vision.VisionTrainer.  LoRA (--lora_enable True, the shipped recipe): lora.LoraTrainer.

Schedule and groups: cosine with linear warm-up over ceil(warmup_ratio * steps) steps (HF get_cosine_schedule_with_warmup), one
multiplier for every group; --vision_lr / --merger_lr give the vision tower / merger their own base rate (sft_trainer.py:50-110);
no weight decay on biases and norm weights.  Row order: a per-epoch permutation from (seed + epoch); batch k of an epoch goes to
rank k mod world (accelerate's default sharding) -- the order itself is HF / accelerate's RNG use and is "parity unpinned".
"""
from __future__ import annotations

import math
from collections import defaultdict

import numpy as np
import torch

from . import dist as vdist
from .host.sft_dataset import IGNORE_INDEX
from .train import Trainer as UpdateEngine
from .train import cosine_lr


def split_supervised(input_ids, labels):
    """(prompt ids, completion ids, completion mask): the row cut at its first supervised token.  Position 0 can never be supervised
    (nothing predicts it); the shifted cross entropy counts exactly the positions >= 1 whose label is not IGNORE_INDEX."""
    ids, lab = np.asarray(input_ids), np.asarray(labels)
    sup = np.nonzero(lab[1:] != IGNORE_INDEX)[0]
    if sup.size == 0:
        return ids, ids[:0], np.zeros(0, dtype=bool)
    p0 = int(sup[0]) + 1
    return ids[:p0], ids[p0:], lab[p0:] != IGNORE_INDEX


class QwenSFTTrainer:
    def __init__(self, model, args, train_dataset, processing_fn, lora=None):
        self.model, self.args, self.dataset, self.processing_fn = model, args, train_dataset, processing_fn
        self.rank, self.world = vdist.rank(), vdist.world_size()
        train_lm = not getattr(args, "freeze_llm", False)
        self.train_tower = not getattr(args, "freeze_vision_tower", True)
        self.train_merger = not getattr(args, "freeze_merger", True)
        vis_tr = None
        if self.train_tower or self.train_merger:
            from .train_vision import VisionTrainer
            vis_tr = VisionTrainer(model.engine, model.vision, self.train_tower, self.train_merger)
        if not (train_lm or vis_tr is not None or lora is not None):
            raise ValueError("nothing to train: --freeze_llm, --freeze_vision_tower and --freeze_merger are all True and LoRA is off")
        self.lora = lora
        self.updater = UpdateEngine(model.engine, model.text, grad_exchange=getattr(args, "grad_exchange_dtype", "fp32"),
                                    vision_trainer=vis_tr, train_lm=train_lm, lora=lora)
        self.bs = max(1, int(args.per_device_train_batch_size))
        self.accum = max(1, int(args.gradient_accumulation_steps))
        per_step = self.bs * self.accum * self.world
        self.steps_per_epoch = len(train_dataset) // per_step
        n_steps = int(math.ceil(self.steps_per_epoch * float(args.num_train_epochs)))
        self.total_steps = args.max_steps if args.max_steps > 0 else max(1, n_steps)
        self.warmup_steps = int(args.warmup_steps) if getattr(args, "warmup_steps", 0) > 0 else int(math.ceil(self.total_steps * args.warmup_ratio))
        self.global_step, self.num_input_tokens_seen = 0, 0
        self.ref_params = None                                   # (the checkpoint helpers shared with the GRPO trainer look at it)
        self.reducer = vdist.MetricsReducer(device=model.device if self.world > 1 and torch.cuda.is_available() else "cpu")
        self._metrics = defaultdict(list)

    # ------------------------------------------------------------------ row order
    def iter_steps(self):
        """yields, per optimiser step, this rank's `accum` micro-batches (lists of `bs` dataset rows); the steps a resumed run
        has already done are skipped"""
        n, per_mb = len(self.dataset), self.bs
        step = 0
        epoch = 0
        while step < self.total_steps:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(int(self.args.seed) + epoch)).tolist()
            batches = [order[i:i + per_mb] for i in range(0, n - per_mb + 1, per_mb)]
            mine = batches[self.rank::self.world][:len(batches) // self.world]
            for s in range(len(mine) // self.accum):
                if step >= self.total_steps:
                    return
                if step >= self.global_step:
                    yield [[self.dataset[i] for i in mb] for mb in mine[s * self.accum:(s + 1) * self.accum]]
                step += 1
            epoch += 1
            if not batches or len(mine) < self.accum:
                return

    # ------------------------------------------------------------------ one optimiser step
    def _row(self, p, scale, last):
        """forward + backward of one processed row; returns (sum of -log p over its supervised tokens [0-d tensor], their count)"""
        eng, dev = self.model.engine, self.model.device
        prompt, comp, mask = split_supervised(p["input_ids"], p["labels"])
        n = int(mask.sum())
        if n == 0:
            return None, 0
        clips = [dict(frames_u8=frames, resize_to=dims) for frames, dims in (p.get("videos") or [])]
        emb = plan = rows = None
        if clips:
            (emb, plan), = eng.encode_videos(clips, per_prompt=[len(clips)])
            if self.updater.vis is not None:
                rows, _ = eng.encode_videos(clips, per_prompt=[len(clips)], rows_only=True)
        comp_d = torch.from_numpy(np.ascontiguousarray(comp[None])).to(dev)
        mask_d = torch.from_numpy(mask[None].astype(np.float32)).to(dev)
        lp = self.updater.forward(prompt, emb, plan, p.get("second_per_grid_ts"), comp_d, 1.0, vision_rows=rows)
        with torch.cuda.stream(eng.stream):
            nll = -(lp * mask_d).sum()
            dlogp = mask_d * (-scale)
        self.updater.backward(dlogp, exchange=last)
        self.num_input_tokens_seen += int(len(p["input_ids"]))
        return nll, n

    def step(self, microbatches, processed=None):
        a, eng = self.args, self.model.engine
        if processed is None:
            processed = [[self.processing_fn(ex) for ex in mb] for mb in microbatches]
        if self.lora is not None:
            self.lora.set_step(self.global_step * self.world + self.rank)       # dropout draws: a function of (seed, step, rank, row)
        mb_losses = []
        for k, mb in enumerate(processed):
            toks = [int(split_supervised(p["input_ids"], p["labels"])[2].sum()) for p in mb]
            n_mb = max(1, sum(toks))
            parts = []
            for j, p in enumerate(mb):
                last = (k == len(processed) - 1) and (j == len(mb) - 1)
                nll, n = self._row(p, 1.0 / (n_mb * len(processed)), last)
                if n:
                    parts.append(nll)
            mb_losses.append((parts, n_mb))
        lr_mult = cosine_lr(1.0, self.global_step, self.total_steps, self.warmup_steps)
        lr = a.learning_rate * lr_mult
        gn = self.updater.optimizer_step(lr=lr, weight_decay=a.weight_decay, beta1=a.adam_beta1, beta2=a.adam_beta2, eps=a.adam_epsilon,
                                         max_grad_norm=a.max_grad_norm,
                                         vision_lr=None if getattr(a, "vision_lr", None) is None else a.vision_lr * lr_mult,
                                         merger_lr=None if getattr(a, "merger_lr", None) is None else a.merger_lr * lr_mult)
        eng.sync()
        loss = float(np.mean([sum(float(x.item()) for x in parts) / n_mb for parts, n_mb in mb_losses])) if mb_losses else float("nan")
        red = self.reducer.reduce({"loss": loss, "grad_norm": float(gn.item()), "learning_rate": lr,
                                   "num_tokens": float(self.num_input_tokens_seen)}, {"num_tokens": "sum"})
        for k_, v in red.items():
            self._metrics[k_].append(v)
        self.global_step += 1
        return dict(loss=red["loss"], metrics=red)

    def log(self):
        out = {k: float(np.mean(v)) for k, v in self._metrics.items() if v}
        if "num_tokens" in self._metrics and self._metrics["num_tokens"]:
            out["num_tokens"] = self._metrics["num_tokens"][-1]
        self._metrics.clear()
        return out

    def non_lora_trainables(self):
        """{HF name: tensor} of the trainable tensors that are not adapters (non_lora_state_dict.bin, train_sft.py:222-229): the
        vision tower / merger when they train beside the adapters; empty for the shipped recipe"""
        sd = self.model.state_dict()
        keep = []
        if self.train_tower:
            keep.append(lambda k: k.startswith("visual.") and not k.startswith("visual.merger."))
        if self.train_merger:
            keep.append(lambda k: k.startswith("visual.merger."))
        # named as peft's wrapper names them (base_model.model.<HF name>): the reference's loader strips that prefix (utils.py:68-71)
        return {"base_model.model." + k: v.cpu() for k, v in sd.items() if any(f(k) for f in keep)}

    # ------------------------------------------------------------------ checkpoints: the GRPO trainer's (same optimiser state layout)
    def save_checkpoint(self, out_dir):
        from .trainer import QwenGRPOTrainer
        return QwenGRPOTrainer.save_checkpoint(self, out_dir)

    def load_checkpoint(self, ckpt_dir):
        from .trainer import QwenGRPOTrainer
        QwenGRPOTrainer.load_checkpoint(self, ckpt_dir)
        return self
