"""GRPO entry point (same path and flags as the reference's train/src/train/train_grpo.py), MI355X engine underneath.

    torchrun --nproc-per-node=N src/train/train_grpo.py --model_id <HF dir> --data_path <json> ... (train/scripts/train_grpo.sh)

What the reference's train() does (/root/reference/train/src/train/train_grpo.py:67-208), step for step:
  parse flags -> load the HF Qwen2_5_VLForConditionalGeneration checkpoint directory (safetensors + config.json) -> dataset
  (video_r2_amd/host/dataset.py) -> reward functions by name from the user's module -> trainer -> train, resuming from the newest
  output_dir/checkpoint-* when there is one (:184-187) -> save the final weights in HF layout (:208).
--save_strategy steps / --save_steps / --save_total_limit behave as HF Trainer's: checkpoint-<step> directories with the
weights (loadable by eval/lmms-eval and demo/ as they are) + optimiser / trainer state, oldest ones pruned.  The frozen KL
reference is always built from --model_id (the SFT model), also on resume.  Sampling seeds are a pure function of
(seed, step, rank) and the prompt order of (seed, epoch), so a resumed run continues the same random streams without
stored RNG state.  The host front end (video decode, frame sampling, overlays, chat template, tokeniser:
video_r2_amd/host/frontend.py) runs --dataloader_num_workers threads one step ahead of the GPU.
"""
import os
import re
import shutil
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
TRAIN_DIR = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))     # so that `src.train.reward_funcs` resolves
for p_ in (ROOT, TRAIN_DIR):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def checkpoints(output_dir):
    """[(step, path)] of output_dir/checkpoint-<step>, oldest first."""
    out = []
    if os.path.isdir(output_dir):
        for name in os.listdir(output_dir):
            m = re.fullmatch(r"checkpoint-(\d+)", name)
            if m and os.path.exists(os.path.join(output_dir, name, "trainer_state.pt")):
                out.append((int(m.group(1)), os.path.join(output_dir, name)))
    return sorted(out)


def prune_checkpoints(output_dir, keep):
    if keep is None or keep <= 0:
        return
    for _, path in checkpoints(output_dir)[:-keep]:
        shutil.rmtree(path, ignore_errors=True)


PROCESSOR_FILES = ("tokenizer.json", "tokenizer_config.json", "vocab.json", "merges.txt", "added_tokens.json", "special_tokens_map.json",
                   "preprocessor_config.json", "video_preprocessor_config.json", "chat_template.json", "chat_template.jinja",
                   "generation_config.json")


def copy_processor_files(model_dir, out_dir):
    """tokeniser / processor / generation configs travel with the weights, so a saved directory loads wherever the input did"""
    if not os.path.isdir(str(model_dir)):
        return
    os.makedirs(out_dir, exist_ok=True)
    for name in PROCESSOR_FILES:
        src = os.path.join(model_dir, name)
        if os.path.exists(src) and not os.path.exists(os.path.join(out_dir, name)):
            shutil.copy2(src, os.path.join(out_dir, name))


class MetricsSink:
    """Where the logged metrics go besides stdout, as HF Trainer does for the reference (`--report_to tensorboard
    --logging_steps 1`, train_grpo.sh:72-75; `trainer.save_state()` train_grpo.py:189): event files under
    output_dir/runs/ when torch.utils.tensorboard imports (it is not in every image), always output_dir/runs/metrics.jsonl, and
    the log history that HF keeps in trainer_state.json (also written into every checkpoint)."""

    def __init__(self, output_dir, report_to, log, resume=""):
        self.history = []
        prev = os.path.join(resume, "trainer_state.json") if resume else ""
        if prev and os.path.exists(prev):          # a resumed run continues the checkpoint's log history, as HF Trainer does
            import json
            self.history = list(json.load(open(prev)).get("log_history", []))
        self.dir = os.path.join(output_dir, "runs")
        os.makedirs(self.dir, exist_ok=True)
        self.jsonl = open(os.path.join(self.dir, "metrics.jsonl"), "a")
        self.tb = None
        if "tensorboard" in str(report_to):
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.tb = SummaryWriter(log_dir=self.dir)
            except Exception as e:        # noqa: BLE001 -- the package is optional; the run must not depend on it
                log(f"--report_to tensorboard: torch.utils.tensorboard is not importable here ({type(e).__name__}); metrics go to "
                    f"{self.dir}/metrics.jsonl and trainer_state.json")

    def write(self, row):
        import json
        self.history.append(row)
        self.jsonl.write(json.dumps(row) + "\n")
        self.jsonl.flush()
        if self.tb is not None:
            for k, v in row.items():
                if k != "step" and isinstance(v, (int, float)):
                    self.tb.add_scalar("train/" + k, v, row["step"])

    def write_completions(self, rows):
        """the prompt / completion / per-function reward rows of a logged step -> output_dir/runs/completions.jsonl"""
        import json
        with open(os.path.join(self.dir, "completions.jsonl"), "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")

    def save_state(self, out_dir, trainer):
        import json
        with open(os.path.join(out_dir, "trainer_state.json"), "w") as f:
            json.dump({"global_step": trainer.global_step, "max_steps": trainer.total_steps,
                       "num_input_tokens_seen": trainer.num_input_tokens_seen, "log_history": self.history}, f, indent=1)

    def close(self):
        self.jsonl.close()
        if self.tb is not None:
            self.tb.close()


def prefetched(steps, processing_fn, workers):
    """(examples, processed) per step; with workers > 0 the host front end of step i + 1 runs on a thread pool while the GPU
    works on step i (decode libraries and PIL release the GIL; frames come back in pinned memory)."""
    if workers <= 0:
        for ex in steps:
            yield ex, [processing_fn(e) for e in ex]
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vr2-frontend") as pool:
        it = iter(steps)
        cur = next(it, None)
        futs = [pool.submit(processing_fn, e) for e in cur] if cur is not None else None
        while cur is not None:
            nxt = next(it, None)
            nfuts = [pool.submit(processing_fn, e) for e in nxt] if nxt is not None else None
            yield cur, [f.result() for f in futs]
            cur, futs = nxt, nfuts


def check_trainable_set(args):
    """Which parameter sets train (train_grpo.py:48-64 of the reference: `configure_vision_tower` sets requires_grad on model.visual
    from --freeze_vision_tower and then on model.visual.merger from --freeze_merger; `configure_llm` on the LM from --freeze_llm).
    The flags' parsed defaults are the reference's: False = trainable.  -> (train_tower, train_merger), also set on `args` as
    `train_vision_tower` / `train_merger` for QwenGRPOTrainer.  Every combination runs: with --freeze_llm True under a trainable vision
    side the LM's backward pass computes input gradients only (the vision gradient arrives through it)."""
    tower, merger = not args.freeze_vision_tower, not args.freeze_merger
    args.train_vision_tower, args.train_merger = tower, merger
    return tower, merger


def train(argv=None, processing_fn=None, dataset=None):
    import torch
    from video_r2_amd import dist as vdist
    from video_r2_amd import params, rewards
    from video_r2_amd.host import dataset as hd
    from video_r2_amd.host import frontend as hfe
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.trainer import QwenGRPOTrainer

    args = params.parse_args(argv)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    rank, world = vdist.init(device=device)
    log = (lambda *a: print("[train_grpo]", *a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)
    if args.ignored:
        log(f"flags accepted and ignored by the MI355X engine: {sorted(args.ignored)}")
    tower, merger = check_trainable_set(args)
    if tower or merger:
        log("trainable besides the LM: " + ", ".join(n for n, on in (("vision tower", tower), ("merger", merger)) if on)
            + " (the KL reference keeps a frozen copy of the vision side)")
    model = VideoR2Model.from_pretrained(args.model_id, device=device)
    lora = None
    if args.lora_enable:                       # train_grpo.py:137-149 of the reference: peft_config -> QwenGRPOTrainer(peft_config=...)
        from video_r2_amd.lora import LoraTrainer
        lora = LoraTrainer(model.engine, model.text, rank=args.lora_rank, alpha=args.lora_alpha, dropout=args.lora_dropout,
                           num_modules=args.num_lora_modules, seed=args.seed)
        log(f"LoRA: {len(lora.module_names())} adapted modules (rank {args.lora_rank}, alpha {args.lora_alpha}, dropout {args.lora_dropout}); the KL "
            f"reference is the adapter-disabled model")
    if dataset is None:
        if not args.data_path:
            raise ValueError("--data_path is required")
        dataset = hd.make_grpo_data_module(args.model_id, None, args)["train_dataset"]
    if processing_fn is None:
        tok = hfe.load_tokenizer(args.model_id if os.path.isdir(str(args.model_id)) else None, model.cfg)
        log(f"tokeniser: {type(tok).__name__}; video front end on {max(0, args.dataloader_num_workers)} prefetch thread(s)")
        processing_fn = hfe.ProcessingFrontEnd(model.cfg, tok, model_dir=args.model_id if os.path.isdir(str(args.model_id)) else None)
    module = rewards.resolve_reward_module(args.reward_module)
    funcs, weights, names = rewards.load_reward_funcs(module, args)
    log(f"reward functions {names} (weights {weights}) from {module}")
    sentence_model = rewards.load_sentence_model(names, args.sentence_model_name)
    reward_judge = rewards.load_reward_judge(names, args.reward_llm_judge)
    trainer = QwenGRPOTrainer(model, funcs, args, dataset, processing_fn, reward_weights=weights, reward_func_names=names,
                              sentence_model=sentence_model, reward_judge=reward_judge, lora=lora)
    resume = args.resume_from_checkpoint or (checkpoints(args.output_dir)[-1][1] if checkpoints(args.output_dir) else "")
    if resume:
        trainer.load_checkpoint(resume)
        log(f"resumed from {resume} at step {trainer.global_step}")
    log(f"{trainer.total_steps} optimiser steps, {trainer.groups_per_step} prompt groups x {args.num_generations} completions each")
    save_every = args.save_steps if (args.save_strategy == "steps" and args.save_steps > 0) else 0
    sink = MetricsSink(args.output_dir, args.report_to, log, resume) if rank == 0 else None
    t0 = time.time()
    for examples, processed in prefetched(trainer.iter_steps(), processing_fn, args.dataloader_num_workers):
        if trainer.global_step >= trainer.total_steps:
            break
        trainer.step(examples, processed)
        step = trainer.global_step
        if rank == 0 and (step % max(1, args.logging_steps) == 0 or step == trainer.total_steps):
            row = {"step": step, "epoch": round(step / max(1, trainer.steps_per_epoch * max(1, args.num_iterations)), 4),
                   "elapsed_s": round(time.time() - t0, 1), **trainer.log()}
            print(row, flush=True)
            sink.write(row)
        if rank == 0 and getattr(trainer, "completions_table", None):     # --log_completions: the step's table (grpo_trainer.py:1206-1227)
            sink.write_completions(trainer.completions_table)
            trainer.completions_table = None
        if save_every and step % save_every == 0:
            if rank == 0:                                   # replicas are bit-identical: one writer
                path = trainer.save_checkpoint(args.output_dir)
                copy_processor_files(args.model_id, path)
                sink.save_state(path, trainer)
                prune_checkpoints(args.output_dir, args.save_total_limit)
                log(f"saved {path}")
            if world > 1:
                import torch.distributed as dist
                dist.barrier()
    if rank == 0:
        if lora is not None:                   # train_grpo.py:193-204: the adapters in peft's layout + every non-adapter tensor
            model.save_config(args.output_dir)
            lora.save_adapter(args.output_dir, base_model_name_or_path=str(args.model_id))
            torch.save({"base_model.model." + k: v.cpu() for k, v in model.state_dict().items()},
                       os.path.join(args.output_dir, "non_lora_state_dict.bin"))
        else:
            model.save_pretrained(args.output_dir)
        copy_processor_files(args.model_id, args.output_dir)
        sink.save_state(args.output_dir, trainer)          # trainer.save_state() of the reference (train_grpo.py:189)
        sink.close()
        log(f"final weights in {args.output_dir}")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    train()
