"""Drop-in model façade: the two calls the reference trainer makes on its HF model, served by the MI355X engine.

  * ``generate(input_ids=, attention_mask=, pixel_values_videos=, video_grid_thw=, second_per_grid_ts=, [pixel_values=,
    image_grid_thw=,] generation_config=)`` -> LongTensor [B, S + C]   (/root/reference/train/src/trainer/grpo_trainer.py:783-786)
  * ``per_token_logps(input_ids, attention_mask, logits_to_keep, **multimodal_inputs)`` -> [B, C]   (grpo_trainer.py:723-741)
  * ``from_pretrained(path)`` / ``save_pretrained(path)`` in HF layout (train_grpo.py:114-119, :208)

The reference batches G left-padded copies of one prompt per group; here identical consecutive rows are recognised as one
prompt group: the ViT and the prompt prefill run once per group and the prefix KV is shared by its rows.  Padding is
stripped (a masked key is never attended and positions are computed on unmasked tokens, so the result is identical).
Raises ValueError when video-token count and feature rows differ (HF modeling_qwen2_5_vl.py:1128-1132).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import weights as W
from .config import ModelConfig
from .engine import Engine


@dataclass
class GenerationConfig:
    max_new_tokens: int = 1024
    do_sample: bool = True
    temperature: float = 0.9
    top_p: float = 1.0
    top_k: int = 50
    min_p: float | None = None
    repetition_penalty: float = 1.0
    pad_token_id: int | None = None
    eos_token_id: int | None = None
    seed: int | None = None      # None: every generate() call draws its own seed from torch's global CPU generator (HF draws its
                                 # samples from the global RNG: torch.manual_seed makes a run repeatable, two calls differ); an int
                                 # pins the call's Philox stream exactly


def _groups(ids, mask, same_video=None):
    """Consecutive rows with identical unpadded ids AND identical video inputs -> [(first_row, n_rows, unpadded_ids)].
    same_video(b) says whether row b carries the same clips (grids and pixel values) as row b - 1: video placeholder ids
    are the same for every clip of one geometry, so equal ids alone do not make two rows copies of one sample."""
    out = []
    prev = None
    for b in range(ids.shape[0]):
        row = ids[b][mask[b].astype(bool)]
        if prev is not None and row.shape == prev.shape and np.array_equal(row, prev) and (same_video is None or same_video(b)):
            out[-1][1] += 1
        else:
            out.append([b, 1, row])
        prev = row
    return out


class _Media:
    """The batch's vision inputs as HF hands them over -- ``pixel_values`` + ``image_grid_thw`` for the still images of ALL rows,
    ``pixel_values_videos`` + ``video_grid_thw`` (+ ``second_per_grid_ts``) for all the clips, each list in batch-then-prompt
    order (what `processor(text=, images=, videos=)` returns, grpo_trainer.py:759-767) -- dealt to the rows the way HF's
    masked_scatter consumes them: row after row, and inside a row one pad-token run after the other."""

    def __init__(self, cfg, ids, mask, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts):
        self.cfg = cfg
        self.px = {"image": pixel_values, "video": pixel_values_videos}
        self.grids, self.offs = {}, {}
        for kind, g in (("image", image_grid_thw), ("video", video_grid_thw)):
            g = np.zeros((0, 3), dtype=np.int64) if g is None else np.asarray(g.cpu() if torch.is_tensor(g) else g).reshape(-1, 3)
            self.grids[kind] = g
            self.offs[kind] = np.concatenate([[0], np.cumsum(g[:, 0] * g[:, 1] * g[:, 2])]).astype(np.int64)
        self.secs = None if second_per_grid_ts is None else [float(x) for x in np.asarray(
            second_per_grid_ts.cpu() if torch.is_tensor(second_per_grid_ts) else second_per_grid_ts).reshape(-1)]
        unit = cfg.vision.spatial_merge_size ** 2
        tok_id = {"image": cfg.image_token_id, "video": cfg.video_token_id}
        cur = {"image": 0, "video": 0}
        self.items = []                                  # per row: [(kind, index into that kind's grid list)] in prompt order
        for b in range(ids.shape[0]):
            row = ids[b][mask[b].astype(bool)]
            typ = np.where(row == tok_id["video"], 2, np.where(row == tok_id["image"], 1, 0))
            edges = np.flatnonzero(np.diff(typ)) + 1
            its = []
            for a, e in zip(np.concatenate([[0], edges]), np.concatenate([edges, [len(typ)]])):
                if a == e or typ[a] == 0:
                    continue
                kind = "video" if typ[a] == 2 else "image"
                if self.px[kind] is None:
                    if kind == "image" and pixel_values_videos is not None:
                        continue                          # stray image-pad ids in a video prompt: plain tokens, as before
                    raise ValueError(f"prompt has {kind} tokens but no {kind} pixel values were given")
                need, tok = int(e - a), 0
                while tok < need and cur[kind] < len(self.grids[kind]):
                    g = self.grids[kind][cur[kind]]
                    tok += int(g[0] * g[1] * g[2]) // unit
                    its.append((kind, cur[kind]))
                    cur[kind] += 1
                if tok != need:
                    raise ValueError(f"{kind.capitalize()} features and {kind} tokens do not match, tokens: {need}, features: {tok}")
            self.items.append(its)
        for kind in ("image", "video"):                  # features nobody asked for: HF's masked_scatter check fails the same way
            if cur[kind] != len(self.grids[kind]) or (self.px[kind] is not None and len(self.grids[kind]) == 0):
                left = int(self.offs[kind][-1] - self.offs[kind][cur[kind]]) // unit if len(self.grids[kind]) else int(self.px[kind].shape[0]) // unit
                raise ValueError(f"{kind.capitalize()} features and {kind} tokens do not match: {left} merged {kind} feature rows are "
                                 f"left over after the batch's {kind} tokens were served (or no {kind} grid was given)")

    def same(self, b):
        """row b carries the same vision inputs (kinds, grids and pixel values) as row b - 1: placeholder ids are the same for
        every item of one geometry, so equal ids alone do not make two rows copies of one sample"""
        ia, ib = self.items[b - 1], self.items[b]
        if len(ia) != len(ib):
            return False
        for (ka, xa), (kb, xb) in zip(ia, ib):
            if ka != kb or not np.array_equal(self.grids[ka][xa], self.grids[kb][xb]):
                return False
            o, px = self.offs[ka], self.px[ka]
            if not torch.equal(px[int(o[xa]):int(o[xa + 1])], px[int(o[xb]):int(o[xb + 1])]):
                return False
        return True

    def clips(self, b):
        """-> (clips for Engine.encode_videos, second_per_grid_ts per item with None marking a still image)"""
        clips, secs = [], []
        for kind, x in self.items[b]:
            o = self.offs[kind]
            clips.append(dict(pixel_values=self.px[kind][int(o[x]):int(o[x + 1])], grid=tuple(int(v) for v in self.grids[kind][x])))
            secs.append(None if kind == "image" else (1.0 if self.secs is None else self.secs[x]))
        return clips, secs


class VideoR2Model:
    def __init__(self, cfg: ModelConfig, vision: W.Params, text: W.Params, device="cuda:0"):
        self.cfg = cfg
        self.engine = Engine(cfg, vision, text, device=device)
        self.text = text
        self.vision = vision
        self.device = torch.device(device)

    # ------------------------------------------------------------------ loading / saving (HF layout)
    @classmethod
    def from_pretrained(cls, path, device="cuda:0", model_base=None, **_ignored):
        """an HF checkpoint directory, or a directory of LoRA adapters over `model_base` (merged on load, as the reference's
        load_pretrained_model does)"""
        cfg = ModelConfig.from_pretrained(path if os.path.exists(os.path.join(path, "config.json")) or model_base is None else model_base)
        sd = W.load_state_dict(path, model_base=model_base)
        return cls(cfg, W.build_vision(cfg, sd, device), W.build_text(cfg, sd, device), device=device)

    @classmethod
    def from_state_dict(cls, cfg, sd, device="cuda:0"):
        sd = W.normalise_keys(sd)
        return cls(cfg, W.build_vision(cfg, sd, device), W.build_text(cfg, sd, device), device=device)

    def reference_copy(self, sd=None):
        """A second, frozen LM parameter set for the KL reference (grpo_trainer.py:329-346)."""
        if sd is not None:
            return W.build_text(self.cfg, W.normalise_keys(sd), self.device)
        return W.clone_params(self.text)

    def state_dict(self):
        return W.export_state_dict(self.cfg, self.vision, self.text)

    def save_config(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.cfg.to_dict(), f, indent=2)

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        self.save_config(path)
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"),
                  metadata={"format": "pt"})

    # ------------------------------------------------------------------ the two hot calls
    def _media(self, ids, mask, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts):
        return _Media(self.cfg, ids, mask, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)

    def _encode_group(self, media, row):
        """vision items of batch row ``row`` -> (merged tokens in window order, plan, second_per_grid_ts with None at still
        images) for Engine.prefill; (None, None, None) for a text-only row"""
        clips, secs = media.clips(row)
        if not clips:
            return None, None, None
        (emb, plan), = self.engine.encode_videos(clips, per_prompt=[len(clips)])
        return emb, plan, secs

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, pixel_values_videos=None, video_grid_thw=None,
                 second_per_grid_ts=None, generation_config: GenerationConfig | None = None, pixel_values=None,
                 image_grid_thw=None, **hf_kwargs):
        """Also takes HF ``generate`` keyword arguments in place of a generation_config -- max_new_tokens, do_sample,
        temperature, top_k, top_p, min_p, repetition_penalty, num_beams, eos_token_id, pad_token_id, use_cache -- which is how
        the evaluation harness calls its model (/root/reference/eval/lmms-eval/lmms_eval/models/qwen2_5_vl.py:326-336): greedy
        when do_sample is false, beams refused (not silently ignored).  Still images (pixel_values, image_grid_thw: the `images=`
        half of the processor's output, which the trainer forwards with **prompt_inputs, grpo_trainer.py:759-786) run through the
        same vision tower as the clips.  top_p / min_p run on the top-k candidate set (1 <= top_k <= 64), the repetition penalty
        over prompt + generated ids, all in HF's order."""
        if generation_config is None:
            gc = GenerationConfig()
            for k in ("max_new_tokens", "do_sample", "top_k", "eos_token_id", "pad_token_id"):
                if hf_kwargs.get(k) is not None:
                    setattr(gc, k, hf_kwargs[k])
            if hf_kwargs.get("temperature") is not None:
                gc.temperature = float(hf_kwargs["temperature"])
            for k in ("top_p", "min_p", "repetition_penalty"):
                if hf_kwargs.get(k) is not None:
                    setattr(gc, k, float(hf_kwargs[k]))
            if "do_sample" in hf_kwargs and not hf_kwargs["do_sample"]:
                gc.top_p, gc.min_p = 1.0, None                   # HF ignores the warpers without sampling
        else:
            # our dataclass or an HF `transformers.GenerationConfig` (what the reference trainer builds, grpo_trainer.py:559-570):
            # read by attribute name into a config of our own -- the caller's object is never modified, HF's has no `seed`
            gc = GenerationConfig()
            for k in ("max_new_tokens", "do_sample", "temperature", "top_p", "top_k", "min_p", "repetition_penalty", "pad_token_id",
                      "eos_token_id", "seed"):
                v = getattr(generation_config, k, None)
                if v is not None:
                    setattr(gc, k, v)
            if getattr(generation_config, "max_new_tokens", None) is None and getattr(generation_config, "max_length", None):
                gc.max_new_tokens = max(1, int(generation_config.max_length) - int(input_ids.shape[1]))   # HF: total length cap
            if (getattr(generation_config, "num_beams", None) or 1) != 1:
                raise ValueError(f"num_beams={generation_config.num_beams}: the engine samples or decodes greedily, it has no beam search")
            if not gc.do_sample:
                gc.top_p, gc.min_p = 1.0, None
        if (hf_kwargs.get("num_beams") or 1) != 1:
            raise ValueError(f"num_beams={hf_kwargs['num_beams']}: the engine samples or decodes greedily, it has no beam search")
        if isinstance(gc.eos_token_id, (list, tuple)):
            gc.eos_token_id = gc.eos_token_id[0]                 # one stop id (the tokenizer's, as the trainer and the harness pass it)
        if hf_kwargs.get("seed") is not None:
            gc.seed = int(hf_kwargs["seed"])
        if gc.seed is None:
            # HF's transformers.GenerationConfig carries no seed and neither do the harness's keyword arguments: one draw from the
            # global CPU generator per call keys the device's counter-based stream, so a prompt revisited in a later epoch gets new
            # rollouts and torch.manual_seed() still reproduces the run (the trainer passes its own per-step seeds to the engine)
            gc.seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        cfg = self.cfg
        ids = input_ids.cpu().numpy()
        mask = np.ones_like(ids) if attention_mask is None else attention_mask.cpu().numpy()
        B, S = ids.shape
        media = self._media(ids, mask, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        pad = cfg.pad_token_id if gc.pad_token_id is None else gc.pad_token_id
        outs = []
        groups = list(_groups(ids, mask, media.same))
        temp = gc.temperature if gc.do_sample else 1.0
        topk = gc.top_k if gc.do_sample else 1
        warp = dict(top_p=gc.top_p if gc.do_sample else 1.0, min_p=gc.min_p if gc.do_sample else None,
                    repetition_penalty=gc.repetition_penalty)    # (the penalty is a processor: HF applies it to greedy decoding too)
        eos = gc.eos_token_id if gc.eos_token_id is not None else cfg.eos_token_id
        rows_max = self.engine.max_decode_rows()      # sequences per decode weight pass (256 on bf16 weights)
        uniform = len(groups) > 1 and len({n for _, n, _ in groups}) == 1 and groups[0][1] * 2 <= rows_max
        if uniform:
            # every prompt has the same number of copies: prefill the groups into shared-stride prefix slots and decode them
            # TOGETHER, up to rows_max sequences per weight pass (the decode is weight-streaming bound)
            n = groups[0][1]
            per = max(1, rows_max // n)
            for c0 in range(0, len(groups), per):
                chunk = groups[c0:c0 + per]
                slots = self.engine.prefix_slots(len(chunk), max(len(g_[2]) for g_ in chunk), tag="pol") if len(chunk) > 1 else [None]
                states = []
                for p_, (row, n_, ids_row) in enumerate(chunk):
                    emb, plan, sec = self._encode_group(media, row)        # the group's copies carry the same inputs: encoded once
                    states.append(self.engine.prefill(self.text, ids_row, emb, plan, sec, tag="pol", prefix_out=slots[p_]))
                toks, _ = self.engine.decode(self.text, states if len(states) > 1 else states[0], n, gc.max_new_tokens,
                                             temperature=temp, top_k=topk, seed=gc.seed + 7919 * c0, eos_token_id=eos, **warp)
                for p_, (row, n_, _) in enumerate(chunk):
                    outs.append((row, n_, toks[p_ * n:(p_ + 1) * n]))
        else:
            for gi, (row, n, ids_row) in enumerate(groups):
                # every row of the batch carries its own copy of the video: encode the first, skip the copies
                emb, plan, sec = self._encode_group(media, row)
                st = self.engine.prefill(self.text, ids_row, emb, plan, sec, tag="pol")
                toks, _ = self.engine.decode(self.text, st, n, gc.max_new_tokens, temperature=temp, top_k=topk,
                                             seed=gc.seed + 7919 * gi, eos_token_id=eos, **warp)
                outs.append((row, n, toks))
        C = max(t.shape[1] for _, _, t in outs)
        result = torch.full((B, S + C), pad, dtype=torch.long, device=self.device)
        result[:, :S] = input_ids.to(self.device)
        for row, n, t in outs:
            result[row:row + n, S:S + t.shape[1]] = t.long()
        return result

    @torch.no_grad()
    def per_token_logps(self, input_ids, attention_mask, logits_to_keep, temperature=0.9, text_params=None, tag="pol",
                        pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None, pixel_values=None,
                        image_grid_thw=None, **_ignored):
        """input_ids [B, P + C] (prompt left-padded, completion right side), attention_mask [B, P + C] ->
        log-probs [B, C] of the last C tokens.  Positions masked out (after EOS) return 0."""
        cfg = self.cfg
        Wt = self.text if text_params is None else text_params
        ids = input_ids.cpu().numpy()
        mask = np.ones_like(ids) if attention_mask is None else attention_mask.cpu().numpy()
        B, T = ids.shape
        C = int(logits_to_keep)
        P = T - C
        media = self._media(ids[:, :P], mask[:, :P], pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        out = torch.zeros((B, C), dtype=torch.float32, device=self.device)
        for gi, (row, n, ids_row) in enumerate(_groups(ids[:, :P], mask[:, :P], media.same)):
            emb, plan, sec = self._encode_group(media, row)
            st = self.engine.prefill(Wt, ids_row, emb, plan, sec, tag=tag)
            comp = torch.from_numpy(ids[row:row + n, P:]).to(self.device)
            lp = self.engine.score(Wt, st, comp, temperature, tag=tag)
            out[row:row + n] = lp * torch.from_numpy(mask[row:row + n, P:]).to(self.device, torch.float32)
        return out

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None,
                pixel_values=None, image_grid_thw=None, text_params=None, **_ignored):
        """HF-style forward for callers that read ``.logits`` -- the reference's ``_get_per_token_logps`` body as it stands
        (``model(input_ids=, attention_mask=, **multimodal_inputs).logits`` then slice / temperature / selective_log_softmax,
        grpo_trainer.py:725-739).  -> an object with ``logits`` f32 [B, T, V] (row t predicts token t + 1; rows at masked
        positions are zero).  Every batch row runs as one causal sequence of its unmasked tokens; ``per_token_logps`` is the
        fast path (shared prompt prefix, no [T, V] tensor) -- this one exists so that unmodified trainer code keeps working."""
        from types import SimpleNamespace
        Wt = self.text if text_params is None else text_params
        ids = input_ids.cpu().numpy()
        mask = np.ones_like(ids) if attention_mask is None else attention_mask.cpu().numpy()
        B, T = ids.shape
        V = self.cfg.text.vocab_size
        media = self._media(ids, mask, pixel_values, image_grid_thw, pixel_values_videos, video_grid_thw, second_per_grid_ts)
        logits = torch.zeros((B, T, V), dtype=torch.float32, device=self.device)
        for b in range(B):
            keep = mask[b].astype(bool)
            row = ids[b][keep]
            if row.size == 0:
                continue
            emb, plan, sec = self._encode_group(media, b)
            lg = torch.empty((row.shape[0], V), dtype=torch.float32, device=self.device)
            self.engine.prefill(Wt, row, emb, plan, sec, tag="fwd", all_logits=lg)
            logits[b, torch.from_numpy(keep).to(self.device)] = lg
        self.engine.sync()
        return SimpleNamespace(logits=logits)

    __call__ = forward
