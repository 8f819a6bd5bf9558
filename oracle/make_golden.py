"""Pin the oracle against HF transformers and write tests/golden/*.npz.

Run HERE (build container, CPU) only:  python -m oracle.make_golden
It imports `transformers` (5.15.0 installed; the reference pins 4.51.1) -- the third-party package
that holds the hot path's arithmetic -- builds `Qwen2_5_VLForConditionalGeneration` from explicit tiny
configs with OUR seeded weights loaded into it, runs it on CPU fp32, asserts the oracle restatement
agrees (1e-5), and stores inputs + HF outputs as small fixtures.  Nothing here travels to the GPU box
except the .npz files; tests never import transformers.

Drift neutralised: legacy->new key mapping; `mm_token_type_ids` supplied explicitly;
second_per_grid_ts exact values {1.0, 2.0} plus one fractional case flagged "5.15 semantics".
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ints, model as om, vision_index as vi  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_model(cfg: om.Cfg, w):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    v, t = cfg.vision, cfg.text
    hc = Qwen2_5_VLConfig(
        vision_config=dict(depth=v.depth, hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                           num_heads=v.num_heads, in_channels=v.in_channels, patch_size=v.patch_size,
                           spatial_merge_size=v.spatial_merge_size, temporal_patch_size=v.temporal_patch_size,
                           tokens_per_second=v.tokens_per_second, window_size=v.window_size,
                           out_hidden_size=v.out_hidden_size, fullatt_block_indexes=list(v.fullatt_block_indexes)),
        text_config=dict(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps,
                         rope_parameters=dict(rope_type="default", rope_theta=t.rope_theta,
                                              mrope_section=list(t.mrope_section)),
                         eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id, bos_token_id=None),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
        tie_word_embeddings=t.tie_word_embeddings,
    )
    hc._attn_implementation = "sdpa"
    m = Qwen2_5_VLForConditionalGeneration(hc).float().eval()
    # legacy (4.51) -> 5.x key mapping, HF:conversion_mapping.py:707-712
    sd = {}
    for k, val in w.items():
        if k.startswith("visual."):
            sd["model." + k] = val
        elif k.startswith("model."):
            sd["model.language_model." + k[len("model."):]] = val
        else:
            sd[k] = val
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "inv_freq" not in k and not (t.tie_word_embeddings and k == "lm_head.weight")]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def synth_prompt(cfg: om.Cfg, n_frames, hw, n_text=7, seed=0, pad_left=0):
    """Synthetic prompt: [pad]*pad_left + 3 text + vision_start + video_pad*Tv + vision_end + n_text text."""
    g = torch.Generator().manual_seed(seed)
    H, W = hw
    frames = torch.randint(0, 256, (n_frames, 3, H, W), generator=g, dtype=torch.uint8).float().numpy()
    px, grid = vi.patchify(vi.normalize_frames(frames), cfg.vision.patch_size, cfg.vision.spatial_merge_size,
                           cfg.vision.temporal_patch_size)
    tv = grid[0] * grid[1] * grid[2] // cfg.vision.spatial_merge_size ** 2
    lo, hi = 10, min(cfg.text.vocab_size, cfg.image_token_id) - 1
    txt = lambda n: torch.randint(lo, hi, (n,), generator=g).tolist()
    ids = txt(3) + [cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id] + txt(n_text)
    mask = [0] * pad_left + [1] * len(ids)
    ids = [cfg.pad_token_id] * pad_left + ids
    return dict(frames=frames, pixel_values_videos=px.astype(np.float32), video_grid_thw=np.asarray([grid], dtype=np.int64),
                input_ids=np.asarray([ids], dtype=np.int64), attention_mask=np.asarray([mask], dtype=np.int64))


def mm_types(cfg, ids):
    ids = torch.as_tensor(ids)
    return ((ids == cfg.video_token_id) * 2 + (ids == cfg.image_token_id) * 1).int()


def case_forward(name, cfg, seed, n_frames, hw, sec, C, pad_left=0, note=""):
    w = om.make_weights(cfg, seed=seed)
    m = hf_model(cfg, w)
    p = synth_prompt(cfg, n_frames, hw, seed=seed + 1, pad_left=pad_left)
    g = torch.Generator().manual_seed(seed + 2)
    comp = torch.randint(10, cfg.image_token_id - 1, (1, C), generator=g)
    ids = torch.cat([torch.from_numpy(p["input_ids"]), comp], 1)
    am = torch.cat([torch.from_numpy(p["attention_mask"]), torch.ones(1, C, dtype=torch.long)], 1)
    px = torch.from_numpy(p["pixel_values_videos"])
    grid = torch.from_numpy(p["video_grid_thw"])
    spg = torch.tensor([sec], dtype=torch.float32)
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=am, pixel_values_videos=px, video_grid_thw=grid,
                second_per_grid_ts=spg, mm_token_type_ids=mm_types(cfg, ids))
        hf_logits = out.logits.float()
        hf_pos, hf_delta = m.model.get_rope_index(ids, mm_types(cfg, ids), video_grid_thw=grid,
                                                  second_per_grid_ts=spg, attention_mask=am)
        vis = m.model.visual(px, grid_thw=grid)
        hf_video_embeds = vis.pooler_output
    # HF per-token logps, exactly as grpo_trainer.py:723-741 does on top of .logits
    T = 0.9
    lg = hf_logits[:, :-1, :][:, -C:] / T
    tgt = ids[:, 1:][:, -C:]
    hf_logps = torch.gather(lg, -1, tgt[..., None]).squeeze(-1) - torch.logsumexp(lg, -1)
    # oracle check
    o_embeds = om.vision_forward(w, cfg, px, p["video_grid_thw"])
    o_logits = om.forward_logits(w, cfg, ids, am.numpy(), px, p["video_grid_thw"], [sec])
    o_pos, o_delta = vi.rope_index(ids.numpy(), am.numpy(), p["video_grid_thw"], [sec],
                                   video_token_id=cfg.video_token_id, image_token_id=cfg.image_token_id,
                                   tokens_per_second=cfg.vision.tokens_per_second)
    o_logps = om.per_token_logps(w, cfg, ids, am.numpy(), C, T, pixel_values_videos=px,
                                 video_grid_thw=p["video_grid_thw"], second_per_grid_ts=[sec])
    valid = am[0].bool()
    e_vis = (o_embeds - hf_video_embeds).abs().max().item()
    e_log = (o_logits[0, valid] - hf_logits[0, valid]).abs().max().item()
    e_lp = (o_logps - hf_logps).abs().max().item()
    pos_ok = bool((torch.from_numpy(o_pos)[:, 0, valid] == hf_pos[:, 0, valid]).all()) and int(o_delta[0, 0]) == int(hf_delta[0, 0])
    print(f"[{name}] vis {e_vis:.2e} logits {e_log:.2e} logps {e_lp:.2e} pos_ok {pos_ok} delta {int(hf_delta[0,0])} {note}")
    assert e_vis < 2e-4 and e_log < 2e-4 and e_lp < 2e-4 and pos_ok, name
    np.savez_compressed(
        os.path.join(GOLD, f"{name}.npz"),
        seed=np.int64(seed), n_frames=np.int64(n_frames), hw=np.asarray(hw), sec=np.float32(sec), C=np.int64(C),
        group=np.int64(cfg.text.num_attention_heads // cfg.text.num_key_value_heads),
        frames_u8=p["frames"].astype(np.uint8), video_grid_thw=p["video_grid_thw"],
        input_ids=ids.numpy(), attention_mask=am.numpy(), temperature=np.float32(T),
        hf_position_ids=hf_pos.numpy(), hf_rope_delta=hf_delta.numpy(),
        hf_video_embeds=hf_video_embeds.numpy().astype(np.float32),
        hf_logps=hf_logps.numpy().astype(np.float32),
        hf_last_logits=hf_logits[0, -1].numpy().astype(np.float32),
        hf_logits_head=hf_logits[0, -C - 1:, :64].numpy().astype(np.float32),
    )
    return m, w, p


def hf_image_pixels(image_u8_hwc):
    """pixel_values / grid of ONE still image from HF's own image processor (the PIL backend: the torchvision one cannot be
    imported here), resizing off: the reference resizes in fetch_image (vision_process.py:374-422) and hands the processor
    a frame that is already a multiple of 28 inside its pixel range."""
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    out = Qwen2VLImageProcessorPil(do_resize=False)(images=[Image.fromarray(image_u8_hwc)], return_tensors="np")
    return out["pixel_values"].astype(np.float32), tuple(int(x) for x in out["image_grid_thw"][0])


def case_forward_images(name, cfg, seed, image_hws, video, C, pad_left=0, note=""):
    """Still images in the prompt (the `images=` half of the reference's processor call, grpo_trainer.py:759-767; dataset rows
    with an "image" key, grpo_dataset.py:113-125), optionally followed by a video: HF forward with pixel_values / image_grid_thw
    (+ pixel_values_videos / video_grid_thw), the oracle restatement checked against it, inputs + HF outputs stored.
    video = (n_frames, (H, W), second_per_grid_t) or None."""
    w = om.make_weights(cfg, seed=seed)
    m = hf_model(cfg, w)
    g = torch.Generator().manual_seed(seed + 1)
    lo, hi = 10, min(cfg.text.vocab_size, cfg.image_token_id) - 1
    txt = lambda n: torch.randint(lo, hi, (n,), generator=g).tolist()
    unit = cfg.vision.spatial_merge_size ** 2
    ids, images, ipx, igrids = txt(3), [], [], []
    for (H, W) in image_hws:
        img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy()
        px, grid = hf_image_pixels(img)
        o_px, o_grid = vi.patchify(vi.normalize_frames(img.transpose(2, 0, 1)[None].astype(np.float32)), cfg.vision.patch_size,
                                   cfg.vision.spatial_merge_size, cfg.vision.temporal_patch_size)
        assert tuple(o_grid) == grid and np.abs(o_px - px).max() < 1e-6, (name, "image patchify", np.abs(o_px - px).max())
        images.append(img); ipx.append(px); igrids.append(grid)
        ids += [cfg.vision_start_token_id] + [cfg.image_token_id] * (grid[0] * grid[1] * grid[2] // unit) + [cfg.vision_end_token_id] + txt(2)
    vpx = vgrid = frames = None
    sec = 1.0
    if video is not None:
        n_frames, (H, W), sec = video
        frames = torch.randint(0, 256, (n_frames, 3, H, W), generator=g, dtype=torch.uint8).float().numpy()
        vpx, vg = vi.patchify(vi.normalize_frames(frames), cfg.vision.patch_size, cfg.vision.spatial_merge_size, cfg.vision.temporal_patch_size)
        vgrid = np.asarray([vg], dtype=np.int64)
        ids += [cfg.vision_start_token_id] + [cfg.video_token_id] * (vg[0] * vg[1] * vg[2] // unit) + [cfg.vision_end_token_id]
    ids += txt(6)
    mask = [0] * pad_left + [1] * len(ids)
    ids = [cfg.pad_token_id] * pad_left + ids
    comp = torch.randint(lo, hi, (1, C), generator=g)
    ids = torch.cat([torch.tensor([ids]), comp], 1)
    am = torch.cat([torch.tensor([mask]), torch.ones(1, C, dtype=torch.long)], 1)
    ipx_all = np.concatenate(ipx, 0)
    igrid = np.asarray(igrids, dtype=np.int64)
    kw = dict(pixel_values=torch.from_numpy(ipx_all), image_grid_thw=torch.from_numpy(igrid))
    if video is not None:
        kw.update(pixel_values_videos=torch.from_numpy(vpx.astype(np.float32)), video_grid_thw=torch.from_numpy(vgrid),
                  second_per_grid_ts=torch.tensor([sec], dtype=torch.float32))
    with torch.no_grad():
        hf_logits = m(input_ids=ids, attention_mask=am, mm_token_type_ids=mm_types(cfg, ids), **kw).logits.float()
        hf_pos, hf_delta = m.model.get_rope_index(ids, mm_types(cfg, ids), image_grid_thw=kw["image_grid_thw"],
                                                  video_grid_thw=kw.get("video_grid_thw"),
                                                  second_per_grid_ts=kw.get("second_per_grid_ts"), attention_mask=am)
        hf_image_embeds = m.model.visual(kw["pixel_values"], grid_thw=kw["image_grid_thw"]).pooler_output
    T = 0.9
    lg = hf_logits[:, :-1, :][:, -C:] / T
    tgt = ids[:, 1:][:, -C:]
    hf_logps = torch.gather(lg, -1, tgt[..., None]).squeeze(-1) - torch.logsumexp(lg, -1)
    mm = dict(pixel_values=kw["pixel_values"], image_grid_thw=igrid, pixel_values_videos=kw.get("pixel_values_videos"),
              video_grid_thw=vgrid, second_per_grid_ts=None if video is None else [sec])
    o_logps = om.per_token_logps(w, cfg, ids, am.numpy(), C, T, **mm)
    o_embeds = om.vision_forward(w, cfg, kw["pixel_values"], igrid)
    o_pos, o_delta = vi.rope_index(ids.numpy(), am.numpy(), vgrid, mm["second_per_grid_ts"], video_token_id=cfg.video_token_id,
                                   image_token_id=cfg.image_token_id, image_grid_thw=igrid,
                                   tokens_per_second=cfg.vision.tokens_per_second)
    valid = am[0].bool()
    e_vis = (o_embeds - hf_image_embeds).abs().max().item()
    e_lp = (o_logps - hf_logps).abs().max().item()
    pos_ok = bool((torch.from_numpy(o_pos)[:, 0, valid] == hf_pos[:, 0, valid]).all()) and int(o_delta[0, 0]) == int(hf_delta[0, 0])
    print(f"[{name}] image embeds {e_vis:.2e} logps {e_lp:.2e} pos_ok {pos_ok} delta {int(hf_delta[0, 0])} {note}")
    assert e_vis < 2e-4 and e_lp < 2e-4 and pos_ok, name
    extra = {}
    if video is not None:
        extra = dict(frames_u8=frames.astype(np.uint8), video_grid_thw=vgrid, sec=np.float32(sec))
    for k, img in enumerate(images):
        extra[f"image{k}_u8"] = img                                             # HWC uint8, already a multiple of 28
    np.savez_compressed(
        os.path.join(GOLD, f"{name}.npz"), seed=np.int64(seed), C=np.int64(C), n_images=np.int64(len(images)),
        group=np.int64(cfg.text.num_attention_heads // cfg.text.num_key_value_heads),
        pixel_values=ipx_all, image_grid_thw=igrid, input_ids=ids.numpy(), attention_mask=am.numpy(), temperature=np.float32(T),
        hf_position_ids=hf_pos.numpy(), hf_rope_delta=hf_delta.numpy(), hf_image_embeds=hf_image_embeds.numpy().astype(np.float32),
        hf_logps=hf_logps.numpy().astype(np.float32), hf_last_logits=hf_logits[0, -1].numpy().astype(np.float32), **extra)


def case_generate(name, cfg, seed, n_frames, hw, G, C):
    """HF generate(do_sample, T=.9, top_k=50) -> token ids; oracle teacher-forced on those ids must reproduce the
    per-step logits (RNG streams differ between HF/torch.multinomial and anything else)."""
    from transformers import GenerationConfig
    w = om.make_weights(cfg, seed=seed)
    m = hf_model(cfg, w)
    p = synth_prompt(cfg, n_frames, hw, seed=seed + 1, pad_left=0)
    ids = torch.from_numpy(p["input_ids"]); am = torch.from_numpy(p["attention_mask"])
    px = torch.from_numpy(p["pixel_values_videos"]); grid = torch.from_numpy(p["video_grid_thw"])
    gc = GenerationConfig(max_new_tokens=C, do_sample=True, temperature=0.9, top_k=50, top_p=1.0,
                          num_return_sequences=G, pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id,
                          output_scores=False, return_dict_in_generate=True, output_logits=True)
    torch.manual_seed(seed + 5)
    with torch.no_grad():
        out = m.generate(input_ids=ids, attention_mask=am, pixel_values_videos=px, video_grid_thw=grid,
                         second_per_grid_ts=torch.tensor([1.0]), mm_token_type_ids=mm_types(cfg, ids),
                         generation_config=gc)
    seqs = out.sequences
    hf_step_logits = torch.stack(out.logits, 1).float()                     # [G, C', V] raw (pre-warp) logits
    Cn = seqs.shape[1] - ids.shape[1]
    comp = seqs[:, ids.shape[1]:]
    o_seq, o_step = om.generate(w, cfg, ids, am, px, p["video_grid_thw"], [1.0], max_new_tokens=Cn,
                                num_return_sequences=G, forced_tokens=comp, return_step_logits=True,
                                eos_token_id=None)
    # compare only while rows are unfinished in HF (finished rows feed pad tokens)
    mask, eos_idx, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    err = 0.0
    for g in range(G):
        upto = int(min(eos_idx[g] + 1, Cn))
        err = max(err, (o_step[g, :upto] - hf_step_logits[g, :upto]).abs().max().item())
    print(f"[{name}] generate teacher-forced step-logit err {err:.2e}; completion lens {mask.sum(1).tolist()}")
    assert err < 2e-4
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), seed=np.int64(seed), n_frames=np.int64(n_frames),
                        hw=np.asarray(hw), G=np.int64(G), frames_u8=p["frames"].astype(np.uint8),
                        group=np.int64(cfg.text.num_attention_heads // cfg.text.num_key_value_heads),
                        video_grid_thw=p["video_grid_thw"], input_ids=ids.numpy(), attention_mask=am.numpy(),
                        sequences=seqs.numpy(), hf_step_logits_head=hf_step_logits[:, :, :64].numpy(),
                        hf_step_lse=torch.logsumexp(hf_step_logits / 0.9, -1).numpy())


def case_sft(name, cfg, seed, n_frames, hw, sec, spans):
    """Stage-1 SFT: HF's own loss for `labels` (ForCausalLMLoss: shift by one, mean over the positions whose label is not -100 --
    the arithmetic of the reference's mixed-modality forward, train/monkey_patch_forward.py:619-631) and HF autograd's gradients of a
    few tensors, against oracle.sft.sft_loss over oracle forward_logits and torch autograd through the oracle.  spans: [(n tokens,
    supervised)] appended after the video prompt (user / answer pieces)."""
    from oracle import sft as osft
    w = om.make_weights(cfg, seed=seed)
    m = hf_model(cfg, w)
    p = synth_prompt(cfg, n_frames, hw, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    pieces = [(p["input_ids"][0].tolist(), False)]
    for n, sup in spans:
        pieces.append((torch.randint(10, cfg.image_token_id - 1, (n,), generator=g).tolist() + ([cfg.eos_token_id] if sup else []), sup))
    ids_np, labels_np = osft.labels_of(pieces)
    ids, labels = torch.from_numpy(ids_np)[None], torch.from_numpy(labels_np)[None]
    px = torch.from_numpy(p["pixel_values_videos"])
    grid = torch.from_numpy(p["video_grid_thw"])
    spg = torch.tensor([sec], dtype=torch.float32)
    m.train(False)
    for q in m.parameters():
        q.requires_grad_(True)
    out = m(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=px, video_grid_thw=grid, second_per_grid_ts=spg,
            mm_token_type_ids=mm_types(cfg, ids), labels=labels)
    hf_loss = out.loss
    hf_loss.backward()
    named = dict(m.named_parameters())
    pick = {"model.norm.weight": "model.language_model.norm.weight", "lm_head.weight": "lm_head.weight",
            "model.layers.0.self_attn.q_proj.bias": "model.language_model.layers.0.self_attn.q_proj.bias",
            "model.layers.1.mlp.down_proj.weight": "model.language_model.layers.1.mlp.down_proj.weight",
            "visual.merger.ln_q.weight": "model.visual.merger.ln_q.weight",
            "visual.blocks.0.attn.qkv.bias": "model.visual.blocks.0.attn.qkv.bias"}
    hf_grads = {k: named[v].grad.detach().clone() for k, v in pick.items()}
    # oracle: the same loss and gradients by autograd through the restatement
    wt = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    o_logits = om.forward_logits(wt, cfg, ids, None, px, p["video_grid_thw"], [sec])
    o_loss = osft.sft_loss(o_logits, labels)
    o_loss.backward()
    e_loss = abs(float(o_loss) - float(hf_loss))
    e_g = {k: ((wt[k].grad - hf_grads[k]).norm() / hf_grads[k].norm().clamp(min=1e-12)).item() for k in pick}
    print(f"[{name}] loss {float(hf_loss):.6f} |oracle - hf| {e_loss:.2e}; gradient rel err {max(e_g.values()):.2e} ({len(ids_np)} tokens, "
          f"{int((labels_np != -100).sum())} supervised)")
    assert e_loss < 1e-5 * max(1.0, abs(float(hf_loss))) and max(e_g.values()) < 1e-4, (e_loss, e_g)
    np.savez_compressed(
        os.path.join(GOLD, f"{name}.npz"), seed=np.int64(seed), n_frames=np.int64(n_frames), hw=np.asarray(hw), sec=np.float32(sec),
        group=np.int64(cfg.text.num_attention_heads // cfg.text.num_key_value_heads), frames_u8=p["frames"].astype(np.uint8),
        video_grid_thw=p["video_grid_thw"], input_ids=ids_np, labels=labels_np, hf_loss=np.float32(float(hf_loss)),
        hf_lm_head_grad_rows=hf_grads["lm_head.weight"][:64].numpy().astype(np.float32),
        **{"hf_grad:" + k: (v[:16] if v.dim() == 2 else v).numpy().astype(np.float32) for k, v in hf_grads.items() if k != "lm_head.weight"})   # (matrices: 16 rows)


def case_indices():
    """Index KATs straight from transformers.vision_utils + torch for several grids."""
    from transformers.vision_utils import get_vision_cu_seqlens, get_vision_position_ids, get_vision_window_index
    out = {}
    grids = [(2, 16, 16), (16, 20, 36), (1, 8, 12), (3, 18, 26), (2, 4, 4), (8, 16, 16), (2, 24, 24)]
    for gi, g in enumerate(grids):
        gt = torch.tensor([g])
        wi, cw = get_vision_window_index(gt, 2, 112, 14)
        out[f"g{gi}_grid"] = np.asarray(g)
        out[f"g{gi}_window_index"] = wi.numpy()
        out[f"g{gi}_cu_window"] = cw.numpy()
        out[f"g{gi}_cu_full"] = get_vision_cu_seqlens(gt).numpy()
        out[f"g{gi}_pos"] = get_vision_position_ids(gt, 2).numpy()
        o_wi, o_cw = vi.vision_window_index([g])
        assert (o_wi == wi.numpy()).all() and (o_cw == cw.numpy()).all(), g
        assert (vi.vision_cu_seqlens([g]) == out[f"g{gi}_cu_full"]).all()
        assert (vi.vision_position_ids([g]) == out[f"g{gi}_pos"]).all()
    # two videos in one batch
    gt = torch.tensor([(2, 8, 8), (4, 12, 20)])
    wi, cw = get_vision_window_index(gt, 2, 112, 14)
    o_wi, o_cw = vi.vision_window_index(gt.tolist())
    assert (o_wi == wi.numpy()).all() and (o_cw == cw.numpy()).all()
    out["multi_grid"] = gt.numpy(); out["multi_window_index"] = wi.numpy(); out["multi_cu_window"] = cw.numpy()
    # linspace frame indices + randperm sampler orders (torch)
    ls = []
    for tot, n in [(300, 32), (299, 32), (17, 4), (1000, 64), (5, 2), (31, 30), (123457, 32), (64, 64), (33, 16)]:
        idx = torch.linspace(0, tot - 1, n).round().long().numpy()
        assert ints.frame_indices(tot, n) == idx.tolist()
        ls.append(np.concatenate([[tot, n], idx]))
    out["linspace_cases"] = np.concatenate(ls)
    for seed, n in [(42, 100), (0, 7), (1234, 1000)]:
        gg = torch.Generator(); gg.manual_seed(seed)
        out[f"randperm_{seed}_{n}"] = torch.randperm(n, generator=gg).numpy()
    np.savez_compressed(os.path.join(GOLD, "indices.npz"), **out)
    print("[indices] ok")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "sft":       # only the SFT fixtures (the others are byte-stable: nothing to regenerate)
        case_sft("sft_tiny_g2", om.cfg_tiny(2), seed=41, n_frames=4, hw=(56, 84), sec=1.0, spans=[(9, True), (5, False), (6, True)])
        return
    case_indices()
    case_forward("fwd_tiny_g2", om.cfg_tiny(2), seed=11, n_frames=4, hw=(112, 112), sec=1.0, C=9)
    case_forward("fwd_tiny_g2_pad", om.cfg_tiny(2), seed=12, n_frames=2, hw=(84, 140), sec=2.0, C=6, pad_left=5,
                 note="(left pad, ragged windows)")
    case_forward("fwd_tiny_g7", om.cfg_tiny(7), seed=13, n_frames=4, hw=(140, 196), sec=1.0, C=8,
                 note="(GQA group 7, ragged windows)")
    case_forward("fwd_tiny_g2_frac", om.cfg_tiny(2), seed=14, n_frames=4, hw=(56, 56), sec=1.5, C=4,
                 note="(fractional second_per_grid_ts: 5.15 semantics int() truncation)")
    case_generate("gen_tiny_g2", om.cfg_tiny(2), seed=21, n_frames=4, hw=(112, 112), G=3, C=12)
    case_forward_images("fwd_tiny_g2_img", om.cfg_tiny(2), seed=31, image_hws=[(84, 112)], video=(4, (56, 84), 1.0), C=7,
                        note="(one still image, then a video)")
    case_forward_images("fwd_tiny_g2_img2", om.cfg_tiny(2), seed=32, image_hws=[(140, 84), (56, 56)], video=None, C=5, pad_left=3,
                        note="(two still images, no video, left pad)")
    case_sft("sft_tiny_g2", om.cfg_tiny(2), seed=41, n_frames=4, hw=(56, 84), sec=1.0, spans=[(9, True), (5, False), (6, True)])


if __name__ == "__main__":
    main()
