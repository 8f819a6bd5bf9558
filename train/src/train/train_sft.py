"""Stage-1 SFT entry point (same path and flags as the reference's train/src/train/train_sft.py), MI355X engine underneath.

    torchrun --nproc-per-node=N src/train/train_sft.py --model_id <HF dir> --data_path <json> ... (train/scripts/train_sft.sh)

What the reference's train() does (/root/reference/train/src/train/train_sft.py:72-231), step for step: parse flags -> check the
LoRA / freeze combinations (:88-108) -> load the HF Qwen2_5_VLForConditionalGeneration checkpoint directory -> requires_grad from
the freeze flags (:41-60,137-139) -> LoRA adapters on the LM's linear layers when --lora_enable (:153-186) -> dataset + collator
(video_r2_amd/host/sft_dataset.py) -> QwenSFTTrainer.train(), resuming from the newest output_dir/checkpoint-* (:209-212) ->
with LoRA: adapter_model.safetensors + adapter_config.json (PEFT layout) and non_lora_state_dict.bin (:218-229); without: the full
weights in HF layout (:231).  `python src/merge_lora_weights.py --model-path <adapter dir> --model-base <base> --save-model-path <out>`
folds the adapters into the base weights for the GRPO stage, as the reference's tool of that name does.
"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
TRAIN_DIR = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p_ in (ROOT, TRAIN_DIR, os.path.dirname(os.path.abspath(__file__))):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

from train_grpo import MetricsSink, checkpoints, copy_processor_files, prefetched, prune_checkpoints   # noqa: E402


def _n_model_params(m):
    """parameter count of the HF module tree (tile padding and the engine's transposed / fused copies do not count)"""
    v, t = m.cfg.vision, m.cfg.text
    hv, iv, mh = v.hidden_size, v.intermediate_size, v.hidden_size * v.spatial_merge_size ** 2
    vis = hv * v.patch_dim + v.depth * (2 * hv + 3 * hv * hv + 3 * hv + hv * hv + hv + 3 * hv * iv + 2 * iv + hv) \
        + hv + mh * mh + mh + mh * v.out_hidden_size + v.out_hidden_size
    h, i = t.hidden_size, t.intermediate_size
    nqkv = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    lm = t.vocab_size * h * (1 if t.tie_word_embeddings else 2) + h \
        + t.num_hidden_layers * (2 * h + nqkv * h + nqkv + t.num_attention_heads * t.head_dim * h + 3 * h * i)
    return vis + lm


def count_parameters(trainer, log):
    """the three lines train_sft.py:62-69 prints"""
    total = _n_model_params(trainer.model)
    up = trainer.updater
    trainable = up.state.n_params() + (up.vis.state.n_params() if up.vis is not None else 0) + (up.lora.n_params() if up.lora is not None else 0)
    log(f"Total parameters: {total + (up.lora.n_params() if up.lora is not None else 0):,}")
    log(f"Trainable parameters: {trainable:,}")
    log(f"Frozen parameters: {total - (trainable - (up.lora.n_params() if up.lora is not None else 0)):,}")


def train(argv=None, processing_fn=None, dataset=None):
    import torch
    from video_r2_amd import dist as vdist
    from video_r2_amd import params
    from video_r2_amd.host import frontend as hfe
    from video_r2_amd.host import sft_dataset as hs
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer

    args = params.parse_sft_args(argv)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    rank, world = vdist.init(device=device)
    log = (lambda *a: print("[train_sft]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    if args.ignored:
        log(f"flags accepted and ignored by the MI355X engine: {sorted(args.ignored)}")
    model = VideoR2Model.from_pretrained(args.model_id, device=device)
    lora = None
    if args.lora_enable:
        from video_r2_amd.lora import LoraTrainer
        lora = LoraTrainer(model.engine, model.text, rank=args.lora_rank, alpha=args.lora_alpha, dropout=args.lora_dropout,
                           num_modules=args.num_lora_modules, seed=args.seed)
        log(f"Adding LoRA to the model... {len(lora.module_names())} lora modules (rank {args.lora_rank}, alpha {args.lora_alpha}, "
            f"dropout {args.lora_dropout})")
    if dataset is None:
        if not args.data_path:
            raise ValueError("--data_path is required")
        dataset = hs.make_supervised_data_module(args.model_id, None, args)["train_dataset"]
    if processing_fn is None:
        model_dir = args.model_id if os.path.isdir(str(args.model_id)) else None
        tok = hfe.load_tokenizer(model_dir, model.cfg)
        log(f"tokeniser: {type(tok).__name__}; video front end on {max(0, args.dataloader_num_workers)} prefetch thread(s)")
        processing_fn = hs.SFTFrontEnd(model.cfg, tok, model_dir=model_dir)
    trainer = QwenSFTTrainer(model, args, dataset, processing_fn, lora=lora)
    count_parameters(trainer, log)
    resume = args.resume_from_checkpoint or (checkpoints(args.output_dir)[-1][1] if checkpoints(args.output_dir) else "")
    if resume:
        trainer.load_checkpoint(resume)
        log(f"resumed from {resume} at step {trainer.global_step}")
    log(f"{trainer.total_steps} optimiser steps of {trainer.accum} micro-batch(es) x {trainer.bs} row(s) x {world} rank(s); "
        f"{trainer.warmup_steps} warm-up step(s)")
    save_every = args.save_steps if (args.save_strategy == "steps" and args.save_steps > 0) else 0
    sink = MetricsSink(args.output_dir, args.report_to, log, resume) if rank == 0 else None
    t0 = time.time()

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, args.dataloader_num_workers), thread_name_prefix="vr2-sft-rows")

    def process_step(mbs):                     # the prefetcher hands over one unit per step: a list of micro-batches; its rows in parallel
        futs = [[pool.submit(processing_fn, ex) for ex in mb] for mb in mbs]
        return [[f.result() for f in mb] for mb in futs]

    for (mbs,), (processed,) in prefetched(([s] for s in trainer.iter_steps()), process_step, min(1, args.dataloader_num_workers)):
        if trainer.global_step >= trainer.total_steps:
            break
        trainer.step(mbs, processed)
        step = trainer.global_step
        if rank == 0 and (step % max(1, args.logging_steps) == 0 or step == trainer.total_steps):
            row = {"step": step, "epoch": round(step / max(1, trainer.steps_per_epoch), 4), "elapsed_s": round(time.time() - t0, 1),
                   **trainer.log()}
            print(row, flush=True)
            sink.write(row)
        if save_every and step % save_every == 0:
            if rank == 0:
                path = trainer.save_checkpoint(args.output_dir)
                copy_processor_files(args.model_id, path)
                sink.save_state(path, trainer)
                prune_checkpoints(args.output_dir, args.save_total_limit)
                log(f"saved {path}")
            if world > 1:
                import torch.distributed as dist
                dist.barrier()
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        if lora is not None:                   # train_sft.py:218-229: adapters in PEFT layout + the trainable non-LoRA tensors
            model.save_config(args.output_dir)
            lora.save_adapter(args.output_dir, base_model_name_or_path=str(args.model_id))
            torch.save(trainer.non_lora_trainables(), os.path.join(args.output_dir, "non_lora_state_dict.bin"))
        else:
            model.save_pretrained(args.output_dir)
        copy_processor_files(args.model_id, args.output_dir)
        sink.save_state(args.output_dir, trainer)
        sink.close()
        log(f"final {'adapters' if lora is not None else 'weights'} in {args.output_dir}")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    train()
