"""GRPO trainer over the MI355X engine: the orchestration half of QwenGRPOTrainer
(/root/reference/train/src/trainer/grpo_trainer.py:747-1178) with the same metric names, buffer logic and reward plugin
contract.  DP unit = one prompt group per rank: rewards and group advantages are local, the only
collectives are the packed metrics all-reduce and the gradient all-reduce (video_r2_amd/dist.py).

One step = rollout (ViT, prefill, shared-prefix decode of G completions) -> reference log-probs -> training forward (policy
log-probs with saved activations) -> rewards -> group advantages -> GRPO loss + d(loss)/d(logp) -> hand-written backward ->
gradient all-reduce -> clip + AdamW (video_r2_amd/train.py).  ``num_iterations == 1`` as in train_grpo.sh: the old-policy
log-probs are the detached current ones (grpo_trainer.py:1108-1110).
"""
from __future__ import annotations

from collections import defaultdict

import numpy as np
import torch

from . import dist as vdist
from . import weights as vw
from .host import grpo as hg
from .host.sampler import RepeatSampler, unique_prompts
from .model import GenerationConfig, VideoR2Model
from .train import Trainer as UpdateEngine
from .train import cosine_lr


def call_reward_funcs(reward_funcs, reward_func_names, examples, texts_per_example, args, sentence_model=None, pool=None, workers=1,
                      timer=None, reward_judge=(None, None)):
    """The reward loop of the reference (grpo_trainer.py:846-892) for whole prompt groups: one row per completion, in group
    order.  Every dataset column except "prompt" / "completion" is repeated per completion and forwarded, plus the three knobs
    of :871-875; the TAR family additionally gets sentence_model= (:882-886), a function named reasoning_reward the local judge
    LM as reward_tokenizer= / reward_model= (:877-881).  None -> NaN.  -> fp32 [rows, n_funcs] (CPU).
    workers > 1 (--reward_workers): every function is called on `workers` contiguous slices of the rows from a thread pool -- the
    shipped functions score row by row, and the judge-backed ones spend their time in HTTP round trips (two per completion,
    up to 10 s each: temporal_grounding_reward.py:311-317, reward_funcs.py:103-137), which then overlap."""
    from .rewards import SENTENCE_MODEL_REWARDS
    prompts, completions, cols = [], [], defaultdict(list)
    for ex, texts in zip(examples, texts_per_example):
        for text in texts:
            prompts.append(ex.get("prompt"))
            completions.append([{"role": "assistant", "content": text}])
            for k, v in ex.items():
                if k not in ("prompt", "completion"):
                    cols[k].append(v)
    kw = dict(cols)
    kw.update(length_reinforcement_buffer=args.length_reinforcement_buffer, buffer_seconds=args.buffer_seconds,
              similarity_threshold=args.similarity_threshold)
    out = torch.zeros(len(prompts), len(reward_funcs))
    n = len(prompts)
    cuts = [n * j // workers for j in range(workers + 1)] if (pool is not None and workers > 1 and n >= 2 * workers) else [0, n]
    per_row = {k for k, v in kw.items() if isinstance(v, list) and len(v) == n}
    for i, (fn, name) in enumerate(zip(reward_funcs, reward_func_names)):
        extra = {"sentence_model": sentence_model} if name in SENTENCE_MODEL_REWARDS else {}
        if name == "reasoning_reward":         # the local judge LM of --reward_llm_judge (grpo_trainer.py:877-881)
            extra = {"reward_tokenizer": reward_judge[0], "reward_model": reward_judge[1]}

        def part(a, b, fn=fn, extra=extra):
            sl = {k: (v[a:b] if k in per_row else v) for k, v in kw.items()}
            return fn(prompts=prompts[a:b], completions=completions[a:b], **extra, **sl)
        import contextlib
        with (timer.host(name) if timer is not None else contextlib.nullcontext()):   # profiling_context(self, reward_func_name), :850
            if len(cuts) == 2:
                vals = part(0, n)
            else:
                futs = [pool.submit(part, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
                vals = [v for f in futs for v in f.result()]
        out[:, i] = torch.tensor([float("nan") if v is None else float(v) for v in vals], dtype=torch.float32)
    if len(prompts) and torch.isnan(out).all(dim=1).any():
        row = int(torch.isnan(out).all(dim=1).nonzero()[0, 0])
        import warnings
        warnings.warn(f"All reward functions returned None for prompt {prompts[row]!r} / completion {completions[row]!r}. "
                      "Please ensure that at least one reward function returns a valid reward.")
    return out


class QwenGRPOTrainer:
    def __init__(self, model: VideoR2Model, reward_funcs, args, train_dataset, processing_fn, reward_weights=None,
                 reward_func_names=None, ref_params=None, sentence_model=None, reward_judge=(None, None), lora=None):
        """processing_fn(example) -> dict(input_ids 1-D int array (unpadded prompt), second_per_grid_ts,
        decode(completion_ids) -> list[str], and the clip(s) as ONE of: videos = [(uint8 frames [n,3,H,W], (H', W'))] (resize
        + normalise + patchify on the device: host/frontend.py ProcessingFrontEnd), video_frames_u8 (already resized),
        or pixel_values_videos + video_grid_thw (host-processed, as the HF processor returns them)).  A still image (dataset
        rows with an "image" key) is an item of `videos` holding ONE frame, with None as its second_per_grid_ts entry.
        The host front end (chat template, frame sampling, overlays) stays Python as in the reference (grpo_trainer.py:755-767)."""
        self.model = model
        self.args = args
        self.reward_funcs = list(reward_funcs)
        self.reward_func_names = list(reward_func_names or [f.__name__ for f in reward_funcs])
        self.reward_weights = torch.tensor(reward_weights or [1.0] * len(self.reward_funcs), dtype=torch.float32)
        self.train_dataset = train_dataset
        self.processing_fn = processing_fn
        self.num_generations = args.num_generations
        self.beta = args.beta
        # LoRA (--lora_enable True, peft_config of grpo_trainer.py:256-269,315-318): the policy is base + adapters; the KL reference is the
        # SAME model with the adapters disabled (:340-343,830-833) = the frozen base weights, no second copy.  The passes that take no
        # gradient (rollouts, old-policy scoring) run on a merged copy W + s B A, refreshed after every optimiser step
        self.lora = lora
        if lora is not None:
            if getattr(args, "decode_weights", "bf16") != "bf16" or getattr(args, "sync_ref_model", False):
                raise NotImplementedError("--lora_enable True with --decode_weights fp8 / --sync_ref_model: not built (the reference model is "
                                          "the adapter-disabled policy; the merged rollout copy is bf16)")
            self.ref_params = None if args.beta == 0.0 else model.text
            self.policy = vw.clone_params(model.text)
            lora.merge_into(self.policy, model.text)
        else:
            self.ref_params = ref_params if (ref_params is not None or args.beta == 0.0) else model.reference_copy()
            self.policy = model.text
        self.epsilon_low = args.epsilon
        self.epsilon_high = args.epsilon_high if args.epsilon_high is not None else args.epsilon
        self._metrics = {"train": defaultdict(list), "eval": defaultdict(list)}
        self.global_step = 0
        self.num_input_tokens_seen = 0
        self.sentence_model = sentence_model
        self.reward_judge = reward_judge          # (tokenizer, causal LM) for a reasoning_reward function, else (None, None)
        # HF generate's warpers / processor the reference forwards (grpo_trainer.py:559-570): applied by the decode sampler
        self.sampling = dict(top_p=getattr(args, "top_p", 1.0), min_p=getattr(args, "min_p", None),
                             repetition_penalty=getattr(args, "repetition_penalty", 1.0))
        if (self.sampling["top_p"] not in (None, 1.0) or self.sampling["min_p"]) and not 1 <= int(args.top_k or 0) <= 64:
            raise ValueError(f"--top_p {args.top_p} / --min_p {args.min_p} run on the top-k candidate set: they need 1 <= --top_k <= 64 "
                             f"(got {args.top_k})")
        self.reducer = vdist.MetricsReducer(device=model.device if vdist.world_size() > 1 and torch.cuda.is_available() else "cpu")
        world = vdist.world_size()
        G = self.num_generations
        # The batch flags count COMPLETIONS per device, as in the reference: per optimiser step a rank works on
        # per_device_train_batch_size x gradient_accumulation_steps completions (grpo_trainer.py:667-675).  At G or more a rank
        # owns whole prompt groups: .. // G groups per rank and `world` times as many per step.
        per_rank_completions = args.per_device_train_batch_size * args.gradient_accumulation_steps
        # A rank that is asked for FEWER completions than one group holds (the shipped recipe: 1 per device on 8 devices, G = 8):
        # the reference's sampler repeats every prompt G times and its loader deals the step's global batch to the ranks in
        # contiguous slices (grpo_trainer.py:73-167, :667-675), so ONE prompt group is spread over G / per_rank ranks, each
        # generating and scoring its slice of the completions; rewards are gathered and the advantages computed over the whole
        # group (:846-981).  Same here when the slices tile the groups exactly: `self.group_ranks` ranks share a prompt group,
        # every one of them runs the vision tower and the prompt prefill for it (as every rank of the reference does), decodes
        # its own `per_rank_completions` sequences with its own seed, and the step takes world / group_ranks prompts -- the
        # reference's batch, not world times it.
        self.group_ranks = 1
        self.local_generations = G
        if per_rank_completions < G and G % per_rank_completions == 0 and world % (G // per_rank_completions) == 0:
            self.group_ranks = G // per_rank_completions
            self.local_generations = per_rank_completions
        self.groups_per_rank = max(1, per_rank_completions // G)
        self.groups_per_step = self.groups_per_rank * world // self.group_ranks
        ref_prompts = max(1, per_rank_completions * world // G)
        if vdist.rank() == 0 and (self.groups_per_step != ref_prompts or per_rank_completions % G not in (0, per_rank_completions)):
            import sys
            print(f"[QwenGRPOTrainer] {self.groups_per_step} prompt groups ({self.groups_per_step * G} completions) per optimiser step: "
                  f"{self.groups_per_rank} per rank on {world} rank(s); the reference would take {ref_prompts} prompt(s) per step with these "
                  "flags (their completions per device neither tile a group nor hold whole groups)", file=sys.stderr)
        # normalisation unit of the token-mean losses / metrics (see vr2_grpo_loss): the reference's per-rank micro-batch
        pdb = args.per_device_train_batch_size
        self.norm_rows = 0
        self.block_groups = 1
        if getattr(args, "loss_norm_unit", "per_device_batch") == "per_device_batch":
            if pdb < G:
                if G % pdb:
                    raise ValueError(f"per_device_train_batch_size ({pdb}) must divide num_generations ({G}) or be a multiple of it")
                self.norm_rows = pdb
            else:
                self.block_groups = min(max(1, pdb // G), self.groups_per_rank)   # one micro-batch spans this many groups
        self.sampler = RepeatSampler(train_dataset, mini_repeat_count=self.num_generations, batch_size=self.groups_per_step,
                                     repeat_count=args.num_iterations, shuffle=args.shuffle_dataset, seed=args.seed)
        self._buffered = None
        if getattr(args, "decode_weights", "bf16") in ("fp8", "fp8w"):
            vw.add_fp8_decode_copies(model.text)
            model.engine.decode_fp8 = True
            model.engine.decode_fp8_act = args.decode_weights == "fp8"     # fp8w: weight-only (bf16 activations)
            if args.decode_weights == "fp8":     # ... and the no-grad LM GEMMs (rollout prefill, reference prefill + log-probs) on
                if self.ref_params is not None:  # fp8 operands too (vr2_gemm_a8w8); the reference model is frozen: quantised once
                    vw.add_fp8_decode_copies(self.ref_params)
                model.engine.gemm_fp8 = True
        from .profiling import PhaseTimer
        self.prof = PhaseTimer("QwenGRPOTrainer", stream=model.engine.stream, enabled=getattr(args, "profile_phases", True))
        self._reward_pool = None                     # background thread for the reward functions (see _rollout_all)
        self._row_pool = None                        # --reward_workers threads that split a reward function's rows
        train_lm = not getattr(args, "freeze_llm", False)
        # the vision side trains when asked (train_grpo.py:48-57: requires_grad of model.visual from freeze_vision_tower, then of
        # model.visual.merger from freeze_merger).  Parsed defaults are the reference's (both False = trainable); a trainer built
        # from arguments that do not mention them at all (tests, embedding callers) keeps the frozen default of the shipped recipe
        self.train_tower = getattr(args, "train_vision_tower", False)
        self.train_merger = getattr(args, "train_merger", False)
        self.update = train_lm or self.train_tower or self.train_merger or lora is not None      # anything to train at all
        self.ref_vision = None
        vis_tr = None
        if self.train_tower or self.train_merger:
            from .train_vision import VisionTrainer
            if self.ref_params is not None:      # the frozen reference model keeps the ORIGINAL tower (its own copy, as the reference's)
                self.ref_vision = vw.clone_params(model.vision)
            vis_tr = VisionTrainer(model.engine, model.vision, self.train_tower, self.train_merger)
        # --freeze_llm True under a trainable vision side (configure_llm, train_grpo.py:59-64): the LM's backward still runs -- the
        # vision gradient arrives through it -- but computes input gradients only
        self.updater = (UpdateEngine(model.engine, model.text, grad_exchange=getattr(args, "grad_exchange_dtype", "fp32"), vision_trainer=vis_tr,
                                     train_lm=train_lm, lora=lora) if self.update else None)
        self.steps_per_epoch = len(train_dataset) // self.groups_per_step
        n_steps = int(self.steps_per_epoch * float(args.num_train_epochs)) * max(1, args.num_iterations)
        self.total_steps = args.max_steps if args.max_steps > 0 else max(1, n_steps)
        self._textual_logs = []

    # ------------------------------------------------------------------ rollout + scoring for this rank's prompt groups
    MAX_DECODE_ROWS = None     # sequences per decode weight pass; None = Engine.max_decode_rows() (256 on bf16 weights, 128 on fp8 copies)

    def _clips(self, pi):
        """the clips of one processed example as Engine.encode_videos wants them"""
        vids = pi.get("videos")
        if vids:                                     # uint8 clips: resize + rescale + normalise + patchify + gather on the device
            return [dict(frames_u8=frames, resize_to=dims) for frames, dims in vids]
        if pi.get("video_frames_u8") is not None:    # already resized
            return [dict(frames_u8=pi["video_frames_u8"], resize_to=None)]
        if pi.get("pixel_values_videos") is not None:   # host-processed, as the HF processor returns them (one row block per grid)
            px = pi["pixel_values_videos"]
            grids = np.asarray(pi["video_grid_thw"].cpu() if torch.is_tensor(pi["video_grid_thw"]) else pi["video_grid_thw"]).reshape(-1, 3)
            out, at = [], 0
            for g in grids:
                n = int(g[0] * g[1] * g[2])
                out.append(dict(pixel_values=px[at:at + n], grid=tuple(int(x) for x in g)))
                at += n
            return out
        return []

    def _rollout_all(self, examples, seed, processed=None):
        """Rollouts of all of this rank's prompt groups.  The groups are encoded + prefilled one by one into shared-stride prefix
        slots and then decoded TOGETHER (up to MAX_DECODE_ROWS sequences per weight pass: the decode is weight-streaming bound,
        so P groups cost little more than one).  The reward functions (HTTP judge, embedding model: up to seconds per sample in
        the reference, temporal_grounding_reward.py:311-355) run in a background thread while the GPU scores the completions
        with the reference model; the results are joined before the advantages.  -> (rollouts, rewards_per_func)"""
        a, m = self.args, self.model
        eng = m.engine
        G = self.local_generations              # sequences THIS rank generates per prompt group (num_generations unless ranks share a group)
        per = max(1, (self.MAX_DECODE_ROWS or eng.max_decode_rows()) // G)
        groups = []
        pending = None
        for c0 in range(0, len(examples), per):
            chunk = examples[c0:c0 + per]
            pis = processed[c0:c0 + per] if processed is not None else [self.processing_fn(ex) for ex in chunk]
            # the reference slices its prompt TENSOR to the last max_prompt_length tokens (grpo_trainer.py:772-776) but generates from
            # the unsliced processor output (:783-786): the slice is a no-op on what the model sees, and a prompt longer than the limit
            # would cut video-pad tokens away from their features in the scoring passes.  The engine always takes the whole prompt; a
            # prompt over the limit is an error here, not a silent truncation.
            idss = [np.asarray(pi["input_ids"]).reshape(-1) for pi in pis]
            for ids_ in idss:
                if a.max_prompt_length and len(ids_) > a.max_prompt_length:
                    raise ValueError(f"prompt of {len(ids_)} tokens exceeds --max_prompt_length {a.max_prompt_length}: raise the limit or lower "
                                     "the frame budget (FPS_MAX_FRAMES, --video_max_pixels); truncating would separate video-pad tokens from their features")
            # the chunk's clips go through the vision tower in ONE pass and its prompts through ONE prefill (full GEMM tiles)
            slots = eng.prefix_slots(len(chunk), max(len(i) for i in idss), tag="pol")
            clips = [self._clips(pi) for pi in pis]
            with self.prof.phase("generate"):          # vision tower + prompt prefill + decode (model.generate, :783-786)
                enc = eng.encode_videos([c for cl in clips for c in cl], per_prompt=[len(cl) for cl in clips])
                encs = [(emb, plan, pi.get("second_per_grid_ts")) for (emb, plan), pi in zip(enc, pis)]
                states = eng.prefill_batch(self.policy, [(ids, emb, plan, sec) for ids, (emb, plan, sec) in zip(idss, encs)], slots, tag="pol")
                toks_all, _ = eng.decode(self.policy, states if len(states) > 1 else states[0], G, a.max_completion_length,
                                         temperature=a.temperature, top_k=a.top_k, seed=seed + 7919 * c0,
                                         eos_token_id=m.cfg.eos_token_id, **self.sampling)
            new = []
            for p_, (pi, ids) in enumerate(zip(pis, idss)):
                toks = toks_all[p_ * G:(p_ + 1) * G]
                mask, eos_idx, any_eos = hg.completion_mask(toks, m.cfg.eos_token_id, a.mask_truncated_completions)
                emb, plan, sec = encs[p_]
                # completions are scored and trained on up to their first EOS only: the rows behind it carry no loss (the mask
                # above) and influence nothing before them, so the scoring / training passes pack the sequences at these lengths
                lens = mask.sum(1).cpu().numpy() if getattr(a, "pack_completions", True) else None
                new.append(dict(prompt_ids=ids, completion_ids=toks, completion_mask=mask, any_eos=any_eos, logps=None, lens=lens,
                                ref_logps=None, decode=pi.get("decode"), emb=emb, plan=plan, sec=sec, _state=states[p_],
                                clips=clips[p_] if self.ref_vision is not None or self.train_tower or self.train_merger else None))
            groups.extend(new)
            last = c0 + per >= len(examples)
            if last:      # every completion exists: start the reward functions, then keep the GPU busy with the scoring passes
                eng.sync()
                if getattr(a, "async_rewards", True):
                    if self._reward_pool is None:
                        from concurrent.futures import ThreadPoolExecutor
                        self._reward_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="vr2-rewards")
                    pending = self._reward_pool.submit(self._rewards, examples, groups)
            with self.prof.phase("_get_per_token_logps"):     # (the reference decorates this method: :722)
                # The log-probs that enter the loss beside the policy's -- the frozen reference model's (:826) and, with the LM
                # frozen, the policy's own -- come from Engine.score_unified: the row space and launches of the training forward,
                # so that identical weights give identical bits (kl == 0, ratio == 1 exactly, as in the reference) at every shape.
                for k_, ro in enumerate(new):
                    if self.beta != 0.0:
                        if self.ref_vision is not None and ro["clips"]:   # the policy's tower trains: the reference sees its own
                            (ro["ref_emb"], _), = eng.encode_videos(ro["clips"], per_prompt=[len(ro["clips"])], vision=self.ref_vision)
                        ro["ref_logps"] = eng.score_unified(self.ref_params, ro["prompt_ids"], ro.get("ref_emb", ro["emb"]), ro["plan"], ro["sec"],
                                                            ro["completion_ids"], a.temperature, tag="ref", lens=ro["lens"]).clone()
                        # (clone: the scoring output is a workspace that the next pass over it reuses)
                    if not self.update:
                        ro["logps"] = eng.score_unified(self.policy, ro["prompt_ids"], ro["emb"], ro["plan"], ro["sec"], ro["completion_ids"],
                                                        a.temperature, tag="ref", lens=ro["lens"], allow_fp8=False).clone()
                    ro.pop("_state")
        rewards = pending.result() if pending is not None else self._rewards(examples, groups)
        return groups, rewards

    def _rewards(self, examples, rollouts):
        G = self.local_generations
        texts = [ro["decode"](ro["completion_ids"]) if ro["decode"] else [""] * G for ro in rollouts]
        for ro, t in zip(rollouts, texts):
            ro["texts"] = t
        workers = max(1, int(getattr(self.args, "reward_workers", 1)))
        if workers > 1 and self._row_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._row_pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="vr2-reward-rows")
        return call_reward_funcs(self.reward_funcs, self.reward_func_names, examples, texts, self.args, self.sentence_model,
                                 pool=self._row_pool, workers=workers, timer=self.prof, reward_judge=self.reward_judge)

    # ------------------------------------------------------------------ one optimisation step (forward half)
    def step(self, step_examples, processed=None):
        """step_examples: the prompt groups of THIS rank for this step (see ``iter_steps``); processed: their processing_fn
        results when a prefetcher already ran the host front end (train_grpo.py).
        Completions are regenerated every ``num_iterations`` steps and re-used in between, with the generation-time policy
        log-probs as the old-policy term of the ratio (grpo_trainer.py:1072-1077, :814-821)."""
        a = self.args
        mode = "train"
        if self.global_step % a.num_iterations == 0 or self._buffered is None:
            rollouts, rewards_per_func = self._rollout_all(step_examples, seed=a.seed * 1000003 + self.global_step * 8191 + vdist.rank() * 131,
                                                           processed=processed)
            if self.group_ranks > 1:
                # ranks share prompt groups: gather the reward rows of all ranks (the reference's `gather(rewards_per_func)`,
                # grpo_trainer.py:894-897; rank order = the global batch order, G consecutive rows = one group), compute the
                # advantages over whole groups, keep this rank's rows (:976-981)
                import torch.distributed as dist
                mine = rewards_per_func.to(torch.float32).contiguous()
                backend_dev = self.model.device if dist.get_backend() == "nccl" else "cpu"
                parts = [torch.empty_like(mine, device=backend_dev) for _ in range(vdist.world_size())]
                dist.all_gather(parts, mine.to(backend_dev))
                glob = hg.group_advantages(torch.cat([p_.cpu() for p_ in parts], 0), self.reward_weights, self.num_generations, loss_type=a.loss_type,
                                           scale_rewards=a.scale_rewards, reward_func_names=self.reward_func_names,
                                           positive_reinforcement=a.positive_reinforcement or None)
                n_loc = mine.shape[0]
                lo = vdist.rank() * n_loc
                adv = {k: (v[lo:lo + n_loc] if (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n_loc * vdist.world_size()) else v)
                       for k, v in glob.items()}
            else:
                adv = hg.group_advantages(rewards_per_func, self.reward_weights, self.num_generations, loss_type=a.loss_type,
                                          scale_rewards=a.scale_rewards, reward_func_names=self.reward_func_names,
                                          positive_reinforcement=a.positive_reinforcement or None)
            if a.num_iterations > 1:
                for ro in rollouts:      # old-policy log-probs, fixed for the iterations that re-use this rollout: the training
                    # forward's own arithmetic (unified row space, bf16 GEMMs also under config 5) -> ratio == 1 at iteration 0
                    ro["old_logps"] = self.model.engine.score_unified(self.policy, ro["prompt_ids"], ro["emb"], ro["plan"], ro["sec"],
                                                                      ro["completion_ids"], a.temperature, tag="ref", lens=ro["lens"],
                                                                      allow_fp8=False).clone()
            self._buffered = (rollouts, adv)
        else:
            rollouts, adv = self._buffered
        dev = self.model.device
        eng = self.model.engine
        losses, metrics, outs = [], defaultdict(list), []
        G = self.local_generations               # rows this rank holds per prompt group
        # a reference micro-batch that spans several groups (per_device_train_batch_size > G) normalises the token-mean losses
        # by ITS token count: group g's share is tokens_g / tokens_block, applied as a scale on that group's loss and gradient
        if getattr(self, "lora", None) is not None:
            self.lora.set_step(self.global_step * vdist.world_size() + vdist.rank())   # dropout draws: a function of (seed, step, rank, group)
        scale = [1.0] * len(rollouts)
        if self.block_groups > 1 and a.loss_type in ("bnpo", "mapo_bnpo"):
            ntok = [float(ro["completion_mask"].sum().item()) for ro in rollouts]
            for b0 in range(0, len(rollouts), self.block_groups):
                blk = range(b0, min(b0 + self.block_groups, len(rollouts)))
                tot = max(sum(ntok[g] for g in blk), 1.0)
                for g in blk:
                    scale[g] = max(ntok[g], 1.0) / tot * len(blk)
        for i, ro in enumerate(rollouts):
            live = ro["lens"] is None or int(ro["lens"].sum()) > 0       # (a group whose completions are all masked out has no rows)
            if self.update and live:      # policy log-probs WITH saved activations (the forward of grpo_trainer.py:1094-1096)
                with self.prof.phase("_get_per_token_logps"):
                    rows = None
                    if self.updater.vis is not None and ro.get("clips"):     # the vision side trains: the tower runs inside forward()
                        rows, _ = eng.encode_videos(ro["clips"], per_prompt=[len(ro["clips"])], rows_only=True)
                    ro["logps"] = self.updater.forward(ro["prompt_ids"], ro["emb"], ro["plan"], ro["sec"], ro["completion_ids"], a.temperature,
                                                       lens=ro["lens"], vision_rows=rows)
            elif self.update:
                ro["logps"] = torch.zeros(ro["completion_ids"].shape, dtype=torch.float32, device=dev)
            with torch.cuda.stream(eng.stream):
                out = hg.grpo_loss(ro["logps"], ro["completion_mask"], adv["advantages"][i * G:(i + 1) * G].to(dev),
                                   ref_logps=ro["ref_logps"], old_logps=ro.get("old_logps"), beta=self.beta, epsilon_low=self.epsilon_low,
                                   epsilon_high=self.epsilon_high, loss_type=a.loss_type,
                                   max_completion_length=a.max_completion_length, norm_rows=self.norm_rows, stream=eng.s)
                if scale[i] != 1.0:
                    out["dlogp"].mul_(scale[i]); out["loss"] = out["loss"] * scale[i]
            if self.update and live:
                with self.prof.phase("backward"):
                    self.updater.backward(out["dlogp"], exchange=(i == len(rollouts) - 1))   # gradients accumulate over the groups
            ro["dlogp"] = out["dlogp"]
            outs.append(out)
        eng.sync()                                                   # ONE host sync for the step's groups: the scalars are read
        for out in outs:                                             # here, not per group (each read would drain the stream)
            losses.append(out["loss"].item() / len(rollouts))       # Trainer divides by the accumulation count
            if self.beta != 0.0:
                metrics["kl"].append(out["kl"].item())
            metrics["clip_ratio/low_mean"].append(out["clip_low"].item())
            metrics["clip_ratio/high_mean"].append(out["clip_high"].item())
            metrics["clip_ratio/region_mean"].append(out["clip_region"].item())
        # ---- metrics with the reference's names (grpo_trainer.py:985-1013, :1159-1177); one packed all-reduce
        lens = torch.cat([ro["completion_mask"].sum(1).float().cpu() for ro in rollouts])
        term = torch.cat([ro["any_eos"].cpu() for ro in rollouts])
        self.num_input_tokens_seen += int(sum(len(ro["prompt_ids"]) * G for ro in rollouts) + lens.sum().item())
        tl = lens[term] if term.any() else torch.zeros(1)
        local = {"loss": sum(losses), "completions/mean_length": lens.mean().item(), "completions/min_length": lens.min().item(),
                 "completions/max_length": lens.max().item(), "completions/clipped_ratio": 1 - term.float().mean().item(),
                 "completions/mean_terminated_length": tl.mean().item(), "completions/min_terminated_length": tl.min().item(),
                 "completions/max_terminated_length": tl.max().item(), "reward": adv["mean_grouped"].mean().item(),
                 "reward_std": adv["std_grouped"].mean().item(), "num_tokens": float(self.num_input_tokens_seen)}
        kinds = {"completions/min_length": "min", "completions/max_length": "max", "completions/min_terminated_length": "min",
                 "completions/max_terminated_length": "max", "num_tokens": "sum"}
        for k, v in metrics.items():
            local[k] = float(np.nanmean(v))
        local["clip_ratio/low_min"] = float(np.nanmin(metrics["clip_ratio/low_mean"]))
        local["clip_ratio/high_max"] = float(np.nanmax(metrics["clip_ratio/high_mean"]))
        kinds.update({"clip_ratio/low_min": "min", "clip_ratio/high_max": "max"})
        # per-function reward mean / std over the GLOBAL batch (nanmean / unbiased nanstd of the gathered column,
        # grpo_trainer.py:1005-1010): reduced exactly from (count, sum, sum of squares)
        for j, name in enumerate(self.reward_func_names):
            col = adv["rewards_per_func"][:, j].double()
            ok = ~torch.isnan(col)
            local[f"_r{j}/n"] = float(ok.sum()); local[f"_r{j}/s"] = float(col[ok].sum()); local[f"_r{j}/ss"] = float((col[ok] ** 2).sum())
            kinds.update({f"_r{j}/n": "sum", f"_r{j}/s": "sum", f"_r{j}/ss": "sum"})
        if self.update:
            lr = cosine_lr(a.learning_rate, self.global_step, self.total_steps, int(a.warmup_ratio * self.total_steps))
            with self.prof.phase("optimizer_step"):
                gn = self.updater.optimizer_step(lr=lr, weight_decay=a.weight_decay, beta1=a.adam_beta1, beta2=a.adam_beta2,
                                                 eps=a.adam_epsilon, max_grad_norm=a.max_grad_norm, grad_scale=1.0 / len(rollouts))
            if self.lora is not None:                              # the next rollout samples from the updated adapters
                self.lora.merge_into(self.policy, self.model.text)
            local["grad_norm"] = gn.item()
            eng.sync()
            local["learning_rate"] = lr
            if self.model.engine.decode_fp8:                      # the next rollout decodes the updated policy
                with torch.cuda.stream(self.model.engine.stream):
                    vw.add_fp8_decode_copies(self.model.text)
        eng.sync()
        if self.prof.enabled:                 # profiling/Time taken: QwenGRPOTrainer.<phase or reward function>: seconds, mean over ranks
            local.update(self.prof.collect(["generate", "_get_per_token_logps"] + (["backward", "optimizer_step"] if self.update else [])
                                           + list(self.reward_func_names)))
        red = self.reducer.reduce(local, kinds)
        for j, name in enumerate(self.reward_func_names):
            n_, s_, ss_ = red.pop(f"_r{j}/n"), red.pop(f"_r{j}/s"), red.pop(f"_r{j}/ss")
            mean = s_ / n_ if n_ > 0 else float("nan")
            red[f"rewards/{name}/mean"] = mean
            red[f"rewards/{name}/std"] = (max(ss_ - n_ * mean * mean, 0.0) / (n_ - 1)) ** 0.5 if n_ > 1 else float("nan")
        for k, v in red.items():
            self._metrics[mode][k].append(v)
        if getattr(a, "log_completions", False) and self.global_step % max(1, a.logging_steps) == 0:
            self._log_completions(step_examples, rollouts, adv)
        self.global_step += 1
        if getattr(a, "sync_ref_model", False) and self.ref_params is not None and self.global_step % max(1, a.ref_model_sync_steps) == 0:
            self.sync_ref_model(a.ref_model_mixup_alpha)
        return dict(loss=red["loss"], rollouts=rollouts, advantages=adv, metrics=red, update=self.update)

    def sync_ref_model(self, alpha):
        """TRL's SyncRefModelCallback (--sync_ref_model, grpo_trainer.py:587-588): ref <- alpha * policy + (1 - alpha) * ref on every
        parameter, every ref_model_sync_steps optimiser steps.  Elementwise, so it runs on the tile-packed bf16 copies as they are
        (the mix in fp32, one rounding); replicas apply it to identical weights and stay identical.  The reference model's fp8
        copies (config 5) are re-derived."""
        eng = self.model.engine
        with torch.cuda.stream(eng.stream):
            for k, r in self.ref_params.t.items():
                p = self.model.text.t.get(k)
                if p is None or p.shape != r.shape or p.dtype != r.dtype:
                    continue
                r.copy_(r.float().mul_(1.0 - alpha).add_(p.float(), alpha=alpha).to(r.dtype))
            if getattr(self.ref_params, "f8", None):
                vw.add_fp8_decode_copies(self.ref_params)

    def _log_completions(self, examples, rollouts, adv):  # noqa: C901
        """The step's prompt / completion / per-function reward table on rank 0 (grpo_trainer.py:1016-1019 collects it, :1206-1227
        prints a sample and, under --report_to wandb with a live run, logs it as the table "completions" -- one column per reward
        function, `wandb_log_unique_prompts` drops repeated prompts).  Kept as `completions_table` for the entry point's sink
        (output_dir/runs/completions.jsonl) and sent to wandb when that package is importable and a run is active."""
        G = self.local_generations
        rpf = adv.get("rewards_per_func")
        rows = []
        for i, (ex, ro) in enumerate(zip(examples, rollouts)):
            q = next((c.get("text") for m in (ex.get("prompt") or []) if isinstance(m.get("content"), list) for c in m["content"] if "text" in c), "")
            for g, text in enumerate(ro.get("texts") or []):
                row = {"step": self.global_step, "prompt": q, "completion": text, "reward": float(adv["rewards"][i * G + g])}
                if rpf is not None:
                    for j, name in enumerate(self.reward_func_names):
                        row[name] = float(rpf[i * G + g, j])
                rows.append(row)
        if vdist.world_size() > 1:
            import torch.distributed as dist
            gathered = [None] * vdist.world_size() if vdist.rank() == 0 else None
            dist.gather_object(rows, gathered, dst=0)
            rows = [r for part in (gathered or []) for r in part]
        if vdist.rank() == 0:
            self._textual_logs = rows
            self.completions_table = rows                      # consumed (and cleared) by whoever writes it out
            for r in rows[:4]:
                print(f"[completions] step {r['step']} reward {r['reward']:.3f} | {r['prompt'][:80]!r} -> {r['completion'][:160]!r}")
            if "wandb" in str(getattr(self.args, "report_to", "")):
                try:
                    import wandb
                    if wandb.run is not None:
                        cols = ["step", "prompt", "completion"] + list(self.reward_func_names)
                        seen, data = set(), []
                        for r in rows:
                            if getattr(self.args, "wandb_log_unique_prompts", False) and r["prompt"] in seen:
                                continue
                            seen.add(r["prompt"])
                            data.append([str(r["step"]), r["prompt"], r["completion"]] + [r.get(n) for n in self.reward_func_names])
                        wandb.log({"completions": wandb.Table(columns=cols, data=data)})
                except ImportError:
                    pass

    def iter_steps(self):
        """Yields, per optimisation step, the dataset examples (one per prompt group) owned by this rank.  One pass of the
        seeded sampler per epoch (the order is a pure function of seed and epoch count, so a resumed run skips ahead by
        recomputing it); each rollout batch is yielded num_iterations times (compute_loss buffer logic, :1067-1083)."""
        per_step = self.groups_per_step
        done = self.global_step
        n = 0
        epochs = float(self.args.num_train_epochs)
        it = iter(self.sampler)                      # NB one RepeatSampler pass = one epoch; its generator keeps advancing
        ep = 0
        while ep < epochs:
            order = list(unique_prompts(it, self.num_generations))
            # the sampler repeats every chunk num_iterations times in a row: yield those repeats as consecutive steps
            limit = len(order) if epochs - ep >= 1 else int(len(order) * (epochs - ep))
            for s0 in range(0, limit - per_step + 1, per_step):
                if n >= done:
                    if self.group_ranks > 1:     # several ranks share one prompt group: rank r works on group r // group_ranks
                        mine = [order[s0 + vdist.rank() // self.group_ranks]]
                    else:
                        mine = vdist.shard_prompts(order[s0:s0 + per_step])
                    yield [self.train_dataset[i] for i in mine]
                n += 1
                if self.args.max_steps > 0 and n >= self.args.max_steps:
                    return
            ep += 1
            it = iter(self.sampler)

    # ------------------------------------------------------------------ checkpoint / resume (train_grpo.py:184-208, --save_steps)
    def save_checkpoint(self, out_dir):
        """HF-layout weights (loadable by eval/lmms-eval and demo/ unchanged) + optimiser / trainer state for resume."""
        import os
        d = os.path.join(out_dir, f"checkpoint-{self.global_step}")
        if getattr(self, "lora", None) is not None:     # a peft model's checkpoint: the adapters (+ their optimiser state below), not the base
            self.model.save_config(d)
            self.lora.save_adapter(d, base_model_name_or_path=str(getattr(self.args, "model_id", "")))
            self.lora.save_training_state(d)
        else:
            self.model.save_pretrained(d)
        state = {"global_step": self.global_step, "num_input_tokens_seen": self.num_input_tokens_seen}
        if self.updater is not None:
            st = self.updater.state
            state.update(step_count=st.step_count, master={k: v.cpu() for k, v in st.master.items()},
                         m={k: v.cpu() for k, v in st.m.items()}, v={k: v.cpu() for k, v in st.v.items()})
            if self.updater.vis is not None:
                vs = self.updater.vis.state
                state["vision"] = dict(master={k: v.cpu() for k, v in vs.master.items()}, m={k: v.cpu() for k, v in vs.m.items()},
                                       v={k: v.cpu() for k, v in vs.v.items()})
        if getattr(self.args, "sync_ref_model", False) and self.ref_params is not None:
            state["ref"] = {k: v.cpu() for k, v in self.ref_params.t.items()}     # the reference is no longer --model_id's weights
        torch.save(state, os.path.join(d, "trainer_state.pt"))
        return d

    def load_checkpoint(self, ckpt_dir):
        """Restore optimiser moments, fp32 masters (and the bf16 compute copies derived from them) and the step counters;
        ``iter_steps`` then skips the steps already done (the sampler order is a pure function of the seed)."""
        import os
        state = torch.load(os.path.join(ckpt_dir, "trainer_state.pt"), map_location="cpu", weights_only=True)
        self.global_step = int(state["global_step"])
        self.num_input_tokens_seen = int(state["num_input_tokens_seen"])
        if self.updater is not None and "master" in state:
            st = self.updater.state
            st.step_count = int(state["step_count"])
            for k in st.master:
                st.master[k].copy_(state["master"][k]); st.m[k].copy_(state["m"][k]); st.v[k].copy_(state["v"][k])
            if self.updater.vis is not None and "vision" in state:
                vs = self.updater.vis.state
                for k in vs.master:
                    vs.master[k].copy_(state["vision"]["master"][k]); vs.m[k].copy_(state["vision"]["m"][k]); vs.v[k].copy_(state["vision"]["v"][k])
            self.updater.refresh_compute_copies()
        if getattr(self, "lora", None) is not None and os.path.exists(os.path.join(ckpt_dir, "lora_state.pt")):
            self.lora.load_training_state(ckpt_dir)
            if getattr(self, "policy", None) is not None and self.policy is not self.model.text:
                self.lora.merge_into(self.policy, self.model.text)
        if "ref" in state and self.ref_params is not None:
            for k, v in state["ref"].items():
                if k in self.ref_params.t:
                    self.ref_params.t[k].copy_(v)
            if getattr(self.ref_params, "f8", None):
                vw.add_fp8_decode_copies(self.ref_params)
        return self

    def log(self):
        out = {k: sum(v) / len(v) for k, v in self._metrics["train"].items() if v}
        self._metrics["train"].clear()
        return out
