"""Qwen2.5-VL forward math (ViT + merger + LM + sampling + per-token log-probs), torch-CPU fp32.

Test infrastructure only (see oracle/__init__.py).  This is the checker the HIP path is compared
with (tolerance 1e-3 relative on token log-probs / GRPO loss, stated in the tests).

The arithmetic lives in HF ``transformers`` (reference pin 4.51.1, requirements.txt:3; call sites
/root/reference/train/src/train/train_grpo.py:114-119, src/trainer/grpo_trainer.py:725-727,783-786).
``HF:`` = /usr/local/lib/python3.10/dist-packages/transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py
(5.15.0).  ``oracle/make_golden.py`` checks this file against that implementation and writes the
golden vectors under tests/golden/.

Weights are a flat dict in the LEGACY (4.51) key layout, the one ``MBZUAI/Video-R2-SFT`` ships:
``visual.*``, ``model.layers.*``, ``model.embed_tokens.weight``, ``model.norm.weight``, ``lm_head.weight``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from . import vision_index as vi


@dataclass
class VisionCfg:
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    out_hidden_size: int = 3584
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: tuple = (7, 15, 23, 31)
    in_channels: int = 3
    tokens_per_second: int = 2


@dataclass
class TextCfg:
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: tuple = (16, 24, 24)
    tie_word_embeddings: bool = False


@dataclass
class Cfg:
    vision: VisionCfg = field(default_factory=VisionCfg)
    text: TextCfg = field(default_factory=TextCfg)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_id: int = 151645
    pad_token_id: int = 151643


def cfg_7b():
    return Cfg()


def cfg_3b():
    return Cfg(vision=VisionCfg(out_hidden_size=2048),
               text=TextCfg(hidden_size=2048, num_hidden_layers=36, num_attention_heads=16, num_key_value_heads=2,
                            intermediate_size=11008, vocab_size=151936, tie_word_embeddings=True))


def cfg_tiny(group=2):
    """Small config with the real head dims (80 vision / 128 LM), ragged MLP width, 2 full-attn blocks."""
    kv = 2 if group == 2 else 1
    heads = kv * group
    return Cfg(vision=VisionCfg(depth=4, hidden_size=160, num_heads=2, intermediate_size=212,
                                out_hidden_size=heads * 128, fullatt_block_indexes=(1, 3)),
               text=TextCfg(hidden_size=heads * 128, num_hidden_layers=2, num_attention_heads=heads,
                            num_key_value_heads=kv, intermediate_size=1152, vocab_size=2048),
               image_token_id=2040, video_token_id=2041, vision_start_token_id=2042, vision_end_token_id=2043,
               eos_token_id=2044, pad_token_id=2045)


def weight_shapes(cfg: Cfg):
    """Legacy-layout tensor names -> shapes (SURVEY.md 8(b) 'Weights')."""
    v, t = cfg.vision, cfg.text
    s = {}
    s["visual.patch_embed.proj.weight"] = (v.hidden_size, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
    for i in range(v.depth):
        p = f"visual.blocks.{i}."
        s[p + "norm1.weight"] = (v.hidden_size,)
        s[p + "norm2.weight"] = (v.hidden_size,)
        s[p + "attn.qkv.weight"] = (3 * v.hidden_size, v.hidden_size)
        s[p + "attn.qkv.bias"] = (3 * v.hidden_size,)
        s[p + "attn.proj.weight"] = (v.hidden_size, v.hidden_size)
        s[p + "attn.proj.bias"] = (v.hidden_size,)
        for n in ("gate_proj", "up_proj"):
            s[p + f"mlp.{n}.weight"] = (v.intermediate_size, v.hidden_size)
            s[p + f"mlp.{n}.bias"] = (v.intermediate_size,)
        s[p + "mlp.down_proj.weight"] = (v.hidden_size, v.intermediate_size)
        s[p + "mlp.down_proj.bias"] = (v.hidden_size,)
    mh = v.hidden_size * v.spatial_merge_size ** 2
    s["visual.merger.ln_q.weight"] = (v.hidden_size,)
    s["visual.merger.mlp.0.weight"] = (mh, mh)
    s["visual.merger.mlp.0.bias"] = (mh,)
    s["visual.merger.mlp.2.weight"] = (v.out_hidden_size, mh)
    s["visual.merger.mlp.2.bias"] = (v.out_hidden_size,)
    hd = t.hidden_size // t.num_attention_heads
    s["model.embed_tokens.weight"] = (t.vocab_size, t.hidden_size)
    for i in range(t.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (t.hidden_size,)
        s[p + "post_attention_layernorm.weight"] = (t.hidden_size,)
        s[p + "self_attn.q_proj.weight"] = (t.num_attention_heads * hd, t.hidden_size)
        s[p + "self_attn.q_proj.bias"] = (t.num_attention_heads * hd,)
        s[p + "self_attn.k_proj.weight"] = (t.num_key_value_heads * hd, t.hidden_size)
        s[p + "self_attn.k_proj.bias"] = (t.num_key_value_heads * hd,)
        s[p + "self_attn.v_proj.weight"] = (t.num_key_value_heads * hd, t.hidden_size)
        s[p + "self_attn.v_proj.bias"] = (t.num_key_value_heads * hd,)
        s[p + "self_attn.o_proj.weight"] = (t.hidden_size, t.num_attention_heads * hd)
        s[p + "mlp.gate_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.up_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.down_proj.weight"] = (t.hidden_size, t.intermediate_size)
    s["model.norm.weight"] = (t.hidden_size,)
    if not t.tie_word_embeddings:
        s["lm_head.weight"] = (t.vocab_size, t.hidden_size)
    return s


def make_weights(cfg: Cfg, seed=0, std=0.02, bf16_round=True, norm_jitter=0.1):
    """Seeded synthetic weights: N(0, std) matrices, biases N(0, std), norm weights 1 + jitter*N(0,1).

    ``bf16_round`` rounds every tensor to bf16 and back so that the fp32 oracle and the bf16 HIP engine
    start from bit-identical parameter VALUES (the comparison then measures arithmetic, not storage).
    """
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shp in weight_shapes(cfg).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("layernorm.weight") \
                or name.endswith("ln_q.weight") or name == "model.norm.weight":
            x = 1.0 + norm_jitter * torch.randn(shp, generator=g)
        else:
            x = std * torch.randn(shp, generator=g)
        if bf16_round:
            x = x.to(torch.bfloat16).to(torch.float32)
        w[name] = x
    return w


# ------------------------------------------------------------------------------------------ pieces
def _ident(x):
    return x


def _bf16(x):
    """Round to bfloat16 and back: one rounding point of the reference's torch_dtype=bfloat16 execution."""
    return x.to(torch.bfloat16).to(torch.float32)


def _rounders(emulate_bf16=False, rounding=None):
    """(rd, ro) for the three precisions this restatement runs in.

    rounding = None / "fp32": the checker proper, no rounding anywhere (rd = ro = identity).
    rounding = "bf16" (= emulate_bf16=True): EVERY op's output rounded to bfloat16 -- the precision the reference runs the model
      in (torch_dtype=bfloat16, bf16 residual stream and log-softmax): rd = ro = round.
    rounding = "operands": the HIP engine's own rounding points and nothing else -- the INPUTS of every matrix
      multiply are bfloat16 (normed activations, q / k / v after the rotation, the unnormalised softmax weights, the attention
      output, the SwiGLU / GELU output, the pixel patches), while accumulation, biases, residual stream, norms, softmax sums,
      rotary arithmetic and the log-softmax stay fp32: rd = identity, ro = round.  Comparing the engine with THIS mode isolates
      everything that is not an operand rounding (tile tails, layouts, masks, k order) at widths and depths where the operand
      roundings alone exceed the 1e-3 bar against fp32."""
    if rounding is None:
        rounding = "bf16" if emulate_bf16 else "fp32"
    if rounding not in ("fp32", "bf16", "operands"):
        raise ValueError(f"rounding must be fp32 | bf16 | operands, got {rounding!r}")
    if rounding == "bf16":
        return _bf16, _bf16
    if rounding == "operands":
        return _ident, _bf16
    return _ident, _ident


def _attend(q, k, v, scale, bias, rd, ro):
    """softmax(q k^T * scale + bias) v over the last two dims.  With operand rounding the matrix unit sees bf16 copies of the
    UNNORMALISED weights p = exp(s - max) and the division by sum(p) (fp32, of the unrounded p) comes after P V, as a
    flash-style kernel does it."""
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    if ro is _ident or rd is not _ident:
        att = rd(torch.nan_to_num(s.softmax(-1)))     # nan: fully-masked (padding) query rows
        return rd(att @ v)
    m = s.amax(-1, keepdim=True)
    p = torch.exp(s - torch.where(torch.isfinite(m), m, torch.zeros_like(m)))
    den = p.sum(-1, keepdim=True)
    return torch.nan_to_num((ro(p) @ v) / den)


def fp8_qdq_rows(x):
    """Per-row e4m3 quantise-dequantise of bf16-rounded activations: what vr2_quant_rows_fp8 + the a8w8 GEMVs see (BASELINE
    config 5 with fp8 activations): scale = amax / 448 over the last dim, round to nearest even, x ~= q * scale."""
    x = _bf16(x.float())
    am = x.abs().amax(dim=-1, keepdim=True)
    inv = torch.where(am > 0, torch.full_like(am, 448.0) / am, torch.ones_like(am))   # true division, as the kernel's 448.0f / amax
    sc = torch.where(am > 0, am / 448.0, torch.ones_like(am))
    return (x * inv).to(torch.float8_e4m3fn).float() * sc


def rms_norm(x, w, eps, rd=_ident):
    """HF:65-79 (variance in fp32, cast back to the input dtype, then times weight)."""
    x = x.float()
    var = x.pow(2).mean(-1, keepdim=True)
    return rd(w * rd(x * torch.rsqrt(var + eps)))


def rotate_half(x):
    """HF:153-157."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def vision_forward(w, cfg: Cfg, pixel_values, grid_thw, return_hidden=False, emulate_bf16=False, rounding=None):
    """HF:408-471 (Qwen2_5_VisionTransformerPretrainedModel.forward).  pixel_values [P, C*tp*ps*ps].
    emulate_bf16: round every op's output to bfloat16 (the precision the reference actually runs the model in,
    /root/reference/train/src/train/train_grpo.py compute_dtype) -- used by tests to measure the reference's OWN deviation
    from this fp32 restatement at depths where bf16 noise exceeds the 1e-3 tolerance.  rounding: see _rounders."""
    rd, ro = _rounders(emulate_bf16, rounding)
    v = cfg.vision
    grid = [tuple(int(a) for a in g) for g in np.asarray(grid_thw).reshape(-1, 3)]
    unit = v.spatial_merge_size ** 2
    hd = v.hidden_size // v.num_heads
    x = rd(ro(pixel_values.float()) @ w["visual.patch_embed.proj.weight"].reshape(v.hidden_size, -1).t())   # HF:115-122
    P = x.shape[0]
    win_idx, cu_win = vi.vision_window_index(grid, v.spatial_merge_size, v.window_size, v.patch_size)
    cu_full = vi.vision_cu_seqlens(grid)
    pos = torch.from_numpy(vi.vision_position_ids(grid, v.spatial_merge_size))                         # [P,2]
    win_idx_t = torch.from_numpy(win_idx)
    x = x.reshape(P // unit, unit, -1)[win_idx_t].reshape(P, -1)                                        # HF:436-439
    dim = hd // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))                   # HF:129-131
    rot = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)                                             # [P, hd/2]
    rot = rot.reshape(P // unit, unit, -1)[win_idx_t].reshape(P, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)                                           # [P,1,hd]
    scale = hd ** -0.5
    for li in range(v.depth):
        p = f"visual.blocks.{li}."
        cu = cu_full if li in v.fullatt_block_indexes else cu_win
        h = ro(rms_norm(x, w[p + "norm1.weight"], 1e-6, rd))
        qkv = rd(h @ w[p + "attn.qkv.weight"].t() + w[p + "attn.qkv.bias"])
        q, k, vv = qkv.reshape(P, 3, v.num_heads, hd).permute(1, 0, 2, 3).unbind(0)                     # [P,H,hd]
        q = ro(rd(q * cos + rotate_half(q) * sin))                                                      # HF:160-171 (fp32 inside)
        k = ro(rd(k * cos + rotate_half(k) * sin))
        vv = ro(vv)
        o = torch.empty_like(q)
        for s in range(len(cu) - 1):                                                                    # HF:266-287
            a, b = int(cu[s]), int(cu[s + 1])
            o[a:b] = _attend(q[a:b].transpose(0, 1), k[a:b].transpose(0, 1), vv[a:b].transpose(0, 1), scale, None, rd, ro).transpose(0, 1)
        x = rd(x + rd(ro(o.reshape(P, -1)) @ w[p + "attn.proj.weight"].t() + w[p + "attn.proj.bias"]))
        h = ro(rms_norm(x, w[p + "norm2.weight"], 1e-6, rd))
        g = rd(F.silu(rd(h @ w[p + "mlp.gate_proj.weight"].t() + w[p + "mlp.gate_proj.bias"])))
        u = rd(h @ w[p + "mlp.up_proj.weight"].t() + w[p + "mlp.up_proj.bias"])
        x = rd(x + rd(ro(rd(g * u)) @ w[p + "mlp.down_proj.weight"].t() + w[p + "mlp.down_proj.bias"]))
    m = ro(rms_norm(x, w["visual.merger.ln_q.weight"], 1e-6, rd)).reshape(P // unit, -1)                # HF:147-150
    m = ro(rd(F.gelu(rd(m @ w["visual.merger.mlp.0.weight"].t() + w["visual.merger.mlp.0.bias"]))))
    m = rd(m @ w["visual.merger.mlp.2.weight"].t() + w["visual.merger.mlp.2.bias"])
    m = m[torch.argsort(win_idx_t)]                                                                     # HF:464-466
    return (m, x) if return_hidden else m


def mrope_cos_sin(cfg: Cfg, position_ids):
    """HF:525-538 + :579-585: per-token cos/sin [B,S,hd] with the t/h/w sections interleaved."""
    t = cfg.text
    hd = t.hidden_size // t.num_attention_heads
    inv_freq = 1.0 / (t.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    freqs = position_ids[..., None].float() * inv_freq                                                  # [3,B,S,hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    sec = list(t.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def lm_forward(w, cfg: Cfg, inputs_embeds, position_ids, attention_mask=None, kv_cache=None, emulate_bf16=False, act_quant=None,
               rounding=None, layer_inputs=None, lora=None):
    """HF:790-872 + decoder layers :641-757.  inputs_embeds [B,S,H]; position_ids [3,B,S];
    attention_mask [B, S_total] over cached+new keys (1 = attend).  kv_cache: list of (k,v) [B,Hkv,T,hd]
    (updated in place when given).  Returns final-normed hidden [B,S,H].  rounding: see _rounders.
    layer_inputs: a list that receives the residual stream entering every layer (tests that check one layer at a time).
    lora: {HF weight name: callable(input) -> the adapter branch's output} (oracle/sft.py: peft's lora.Linear.forward adds
    lora_B(lora_A(dropout(x))) * scaling to the base layer's result)."""
    rd, ro = _rounders(emulate_bf16, rounding)

    def lin(h_, name):
        y = h_ @ w[name].t()
        return y + lora[name](h_) if (lora is not None and name in lora) else y

    aq = act_quant if act_quant is not None else _ident       # applied to the input of every projection (config 5, fp8 activations)
    t = cfg.text
    B, S, H = inputs_embeds.shape
    nh, nkv = t.num_attention_heads, t.num_key_value_heads
    hd = H // nh
    cos, sin = mrope_cos_sin(cfg, position_ids)
    cos, sin = rd(cos.unsqueeze(1)), rd(sin.unsqueeze(1))                                               # HF: cos.to(dtype=x.dtype)
    past = 0 if not kv_cache or kv_cache[0] is None else kv_cache[0][0].shape[2]
    T = past + S
    qpos = torch.arange(past, T)
    causal = (torch.arange(T)[None, :] <= qpos[:, None])                                               # [S,T]
    allow = causal[None].expand(B, S, T)
    if attention_mask is not None:
        allow = allow & (attention_mask[:, None, :T] != 0)
    bias = torch.zeros(B, 1, S, T).masked_fill(~allow[:, None], float("-inf"))
    x = rd(inputs_embeds.float())
    for li in range(t.num_hidden_layers):
        p = f"model.layers.{li}."
        if layer_inputs is not None:
            layer_inputs.append(x.clone())
        h = aq(ro(rms_norm(x, w[p + "input_layernorm.weight"], t.rms_norm_eps, rd)))
        q = rd(lin(h, p + "self_attn.q_proj.weight") + w[p + "self_attn.q_proj.bias"]).view(B, S, nh, hd).transpose(1, 2)
        k = rd(lin(h, p + "self_attn.k_proj.weight") + w[p + "self_attn.k_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        v = rd(lin(h, p + "self_attn.v_proj.weight") + w[p + "self_attn.v_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        q = ro(rd(rd(q * cos) + rd(rotate_half(q) * sin)))
        k = ro(rd(rd(k * cos) + rd(rotate_half(k) * sin)))
        v = ro(v)
        if kv_cache is not None:
            if kv_cache[li] is not None:
                k = torch.cat([kv_cache[li][0], k], dim=2)
                v = torch.cat([kv_cache[li][1], v], dim=2)
            kv_cache[li] = (k, v)
        kr = k.repeat_interleave(nh // nkv, dim=1)
        vr = v.repeat_interleave(nh // nkv, dim=1)
        o = _attend(q, kr, vr, hd ** -0.5, bias, rd, ro).transpose(1, 2).reshape(B, S, nh * hd)
        x = rd(x + rd(lin(aq(ro(o)), p + "self_attn.o_proj.weight")))
        h = aq(ro(rms_norm(x, w[p + "post_attention_layernorm.weight"], t.rms_norm_eps, rd)))
        g = rd(rd(F.silu(rd(lin(h, p + "mlp.gate_proj.weight")))) * rd(lin(h, p + "mlp.up_proj.weight")))
        x = rd(x + rd(lin(aq(ro(g)), p + "mlp.down_proj.weight")))
    return ro(rms_norm(x, w["model.norm.weight"], t.rms_norm_eps, rd))


def lm_head_weight(w, cfg):
    return w["model.embed_tokens.weight"] if cfg.text.tie_word_embeddings else w["lm_head.weight"]


def build_inputs_embeds(w, cfg: Cfg, input_ids, video_embeds, image_embeds=None):
    """HF:1206-1223: embedding lookup + masked_scatter of the merged image tokens (image pad positions), then of the merged
    video tokens (video pad positions)."""
    ids = torch.as_tensor(input_ids)
    emb = w["model.embed_tokens.weight"][ids].clone()
    if image_embeds is not None:
        imask = ids == cfg.image_token_id
        if int(imask.sum()) != image_embeds.shape[0]:
            raise ValueError(f"Image features and image tokens do not match, tokens: {int(imask.sum())}, "
                             f"features: {image_embeds.shape[0]}")
        emb[imask] = image_embeds.to(emb.dtype)
    mask = ids == cfg.video_token_id
    if video_embeds is not None:
        if int(mask.sum()) != video_embeds.shape[0]:
            raise ValueError(f"Video features and video tokens do not match, tokens: {int(mask.sum())}, "
                             f"features: {video_embeds.shape[0]}")                                     # HF:1128-1132
        emb[mask] = video_embeds.to(emb.dtype)
    return emb


def forward_logits(w, cfg: Cfg, input_ids, attention_mask, pixel_values_videos, video_grid_thw, second_per_grid_ts,
                   video_embeds=None, emulate_bf16=False, pixel_values=None, image_grid_thw=None, image_embeds=None, rounding=None,
                   lora=None):
    """Full forward -> logits [B,S,V] fp32 (HF:1308-1402 without the loss).  Still images (pixel_values, image_grid_thw: the
    `images=` half of the reference's processor call, grpo_trainer.py:759-767) go through the same vision tower in a pass of
    their own, as HF's get_image_features does."""
    ids = torch.as_tensor(input_ids)
    if video_embeds is None and pixel_values_videos is not None:
        video_embeds = vision_forward(w, cfg, pixel_values_videos, video_grid_thw, emulate_bf16=emulate_bf16, rounding=rounding)
    if image_embeds is None and pixel_values is not None:
        image_embeds = vision_forward(w, cfg, pixel_values, image_grid_thw, emulate_bf16=emulate_bf16, rounding=rounding)
    emb = build_inputs_embeds(w, cfg, ids, video_embeds, image_embeds)
    am = None if attention_mask is None else np.asarray(attention_mask)
    pos, _ = vi.rope_index(ids.numpy(), am, video_grid_thw, second_per_grid_ts, video_token_id=cfg.video_token_id,
                           image_token_id=cfg.image_token_id, image_grid_thw=image_grid_thw,
                           merge_size=cfg.vision.spatial_merge_size, tokens_per_second=cfg.vision.tokens_per_second)
    h = lm_forward(w, cfg, emb, torch.from_numpy(pos), None if attention_mask is None else torch.as_tensor(attention_mask),
                   emulate_bf16=emulate_bf16, rounding=rounding, lora=lora)
    logits = h @ lm_head_weight(w, cfg).t()
    return _rounders(emulate_bf16, rounding)[0](logits)


def selective_log_softmax(logits, index):
    """trl.selective_log_softmax, fp32 branch: gather - logsumexp."""
    sel = torch.gather(logits, -1, index.unsqueeze(-1)).squeeze(-1)
    return sel - torch.logsumexp(logits, dim=-1)


def per_token_logps(w, cfg, input_ids, attention_mask, logits_to_keep, temperature, **mm):
    """/root/reference/train/src/trainer/grpo_trainer.py:723-741."""
    ids = torch.as_tensor(input_ids)
    rounding = mm.get("rounding") or ("bf16" if mm.get("emulate_bf16") else "fp32")
    emu = rounding == "bf16"
    logits = forward_logits(w, cfg, ids, attention_mask, mm.get("pixel_values_videos"), mm.get("video_grid_thw"),
                            mm.get("second_per_grid_ts"), video_embeds=mm.get("video_embeds"), rounding=rounding,
                            pixel_values=mm.get("pixel_values"), image_grid_thw=mm.get("image_grid_thw"),
                            image_embeds=mm.get("image_embeds"), lora=mm.get("lora"))
    logits = logits[:, :-1, :]
    tgt = ids[:, 1:]
    logits = logits[:, -logits_to_keep:]
    tgt = tgt[:, -logits_to_keep:]
    logits = logits / temperature
    if emu:
        # trl.selective_log_softmax, bf16 branch: per-row log_softmax in the logits dtype, then gather
        lsm = _bf16(torch.log_softmax(_bf16(logits), dim=-1))
        return torch.gather(lsm, -1, tgt.unsqueeze(-1)).squeeze(-1)
    return selective_log_softmax(logits, tgt)


def warp_logits(logits, temperature, top_k, top_p=1.0, min_p=None):
    """HF:generation/logits_process.py: temperature (:238), top-k (:542: keep scores >= k-th largest), then -- in HF's warper order --
    TopPLogitsWarper (ascending sort, drop tokens whose cumulative probability is <= 1 - top_p, keep at least one) and
    MinPLogitsWarper (drop tokens with probability < min_p * max probability, keep at least one)."""
    s = logits / temperature
    if top_k and top_k > 0:
        k = min(top_k, s.shape[-1])
        kth = torch.topk(s, k)[0][..., -1, None]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(s, descending=False, dim=-1)
        cum = srt.softmax(-1).cumsum(-1)
        rm = cum <= (1 - top_p)
        rm[..., -1:] = False
        s = s.masked_fill(rm.scatter(-1, idx, rm), float("-inf"))
    if min_p is not None and min_p > 0.0:
        pr = s.softmax(-1)
        rm = pr < min_p * pr.max(-1, keepdim=True)[0]
        top = s.argmax(-1, keepdim=True)
        rm.scatter_(-1, top, False)
        s = s.masked_fill(rm, float("-inf"))
    return s


def repetition_penalty(logits, seen_ids, penalty):
    """HF RepetitionPenaltyLogitsProcessor: for the ids of the sequence so far (row-wise), score < 0 ? score * penalty : score / penalty;
    an id that occurs several times is penalised once (gather -> modify -> scatter)."""
    ids = torch.as_tensor(seen_ids, dtype=torch.long)
    sc = torch.gather(logits, -1, ids)
    sc = torch.where(sc < 0, sc * penalty, sc / penalty)
    return logits.scatter(-1, ids, sc)


def generate(w, cfg: Cfg, input_ids, attention_mask, pixel_values_videos, video_grid_thw, second_per_grid_ts,
             max_new_tokens, temperature=0.9, top_k=50, num_return_sequences=1, generator=None, eos_token_id=None,
             forced_tokens=None, return_step_logits=False):
    """HF generate/_sample loop (HF:generation/utils.py:2783-2937) for ONE left-padded prompt batch.

    Finished rows keep stepping with pad_token_id appended (:2928-2929).  ``forced_tokens`` [B*n, C]
    replaces sampling (teacher forcing) so the HIP path can be compared step for step.
    """
    ids = torch.as_tensor(input_ids)
    B, S = ids.shape
    n = num_return_sequences
    am = torch.ones_like(ids) if attention_mask is None else torch.as_tensor(attention_mask)
    video_embeds = vision_forward(w, cfg, pixel_values_videos, video_grid_thw) if pixel_values_videos is not None else None
    emb = build_inputs_embeds(w, cfg, ids, video_embeds)
    pos, deltas = vi.rope_index(ids.numpy(), am.numpy(), video_grid_thw, second_per_grid_ts,
                                video_token_id=cfg.video_token_id, image_token_id=cfg.image_token_id,
                                merge_size=cfg.vision.spatial_merge_size, tokens_per_second=cfg.vision.tokens_per_second)
    # expand for num_return_sequences (HF repeat_interleave)
    emb = emb.repeat_interleave(n, 0)
    am = am.repeat_interleave(n, 0)
    pos_t = torch.from_numpy(pos).repeat_interleave(n, 1)
    deltas_t = torch.from_numpy(deltas).repeat_interleave(n, 0)                                        # [Bn,1]
    cache = [None] * cfg.text.num_hidden_layers
    h = lm_forward(w, cfg, emb, pos_t, am, cache)
    head = lm_head_weight(w, cfg)
    out = ids.repeat_interleave(n, 0)
    unfinished = torch.ones(B * n, dtype=torch.bool)
    step_logits = []
    last = h[:, -1]
    for step in range(max_new_tokens):
        logits = last @ head.t()
        if return_step_logits:
            step_logits.append(logits.clone())
        if forced_tokens is not None:
            nxt = torch.as_tensor(forced_tokens)[:, step]
        else:
            probs = warp_logits(logits, temperature, top_k).softmax(-1)
            nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
        if eos_token_id is not None:
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, cfg.pad_token_id))
            unfinished = unfinished & (nxt != eos_token_id)
        out = torch.cat([out, nxt[:, None]], dim=1)
        am = torch.cat([am, torch.ones(B * n, 1, dtype=am.dtype)], dim=1)
        if step == max_new_tokens - 1 or (eos_token_id is not None and not unfinished.any()):
            break
        # decode positions: text position (cumsum of mask - 1) + rope_delta   (HF:1164-1181)
        p1 = (am.long().sum(-1, keepdim=True) - 1) + deltas_t                                            # [Bn,1]
        pos_step = p1[None].expand(3, -1, -1)
        e = w["model.embed_tokens.weight"][nxt][:, None]
        last = lm_forward(w, cfg, e, pos_step, am, cache)[:, -1]
    if return_step_logits:
        return out, torch.stack(step_logits, 1)
    return out
