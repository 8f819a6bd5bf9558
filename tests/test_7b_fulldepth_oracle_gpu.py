"""GPU: the FULL-DEPTH Qwen2.5-VL-7B (28 LM layers, 32 ViT blocks, 8.3 B random-init parameters) against the CPU oracle itself --
4 frames of 224^2 (S = 192), G = 2, C = 8: small enough in tokens for the host cores to run 7B in fp32, full size in every
width and in depth.  test_7b_width_oracle_gpu.py holds the widths at 4 layers, test_config1_3b_gpu.py 36 layers of the 3B
widths; test_fullsize_7b_gpu.py is cross-path only.  This case is the one oracle line at 7B depth
(/root/reference/train/src/trainer/grpo_trainer.py:723-741 = what is compared).

Needs ~45 GB of host memory for the fp32 weights (+ the engine's upload copies): skipped cleanly on a smaller host.
Slow (~4-6 min: 8.3 B seeded normal draws, then two fp32 passes of the model on the host)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg

pytestmark = pytest.mark.gpu

NEED_GB = 70


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_7b_28_layers_logps_vs_oracle():
    have = _mem_available_gb()
    if have < NEED_GB:
        pytest.skip(f"host has {have:.0f} GB available; the fp32 7B oracle wants {NEED_GB} GB")
    from video_r2_amd import weights as W
    from video_r2_amd.engine import Engine
    from video_r2_amd.host.processor import video_prompt_ids
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ocfg = om.cfg_7b()
    w = om.make_weights(ocfg, seed=21)                                  # bf16-representable fp32 values, 33 GB on the host
    cfg = to_engine_cfg(ocfg)
    dev = "cuda:0"
    vis = W.build_vision(cfg, w, dev)
    txt = W.build_text(cfg, w, dev)
    eng = Engine(cfg, vis, txt, device=dev)
    g = torch.Generator().manual_seed(23)
    frames = torch.randint(0, 256, (4, 3, 224, 224), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    assert grid == (2, 16, 16)
    txt_ids = torch.randint(1000, 150000, (62,), generator=g).tolist()
    prompt = np.asarray(video_prompt_ids(cfg, 128, txt_ids[:8], txt_ids[8:]))       # S = 128 + 64
    G, C = 2, 8
    emb, plan = eng.encode_video(torch.from_numpy(px).to(dev), [grid])
    emb = emb.clone()
    st = eng.prefill(txt, prompt, emb, plan, [1.0])
    toks, _ = eng.decode(txt, st, G, C, temperature=0.9, top_k=50, seed=29)
    toks = torch.clamp(toks, max=cfg.image_token_id - 1)
    lp = eng.score(txt, st, toks, 0.9).cpu().numpy()
    eng.sync()
    comp = toks.cpu().numpy().astype(np.int64)
    rows = np.concatenate([np.repeat(prompt[None], G, 0), comp], 1)
    mm = dict(video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    ids, am = torch.from_numpy(rows), np.ones_like(rows)
    nat = torch.empty_like(emb)
    nat[torch.from_numpy(plan.window_index).to(emb.device)] = emb
    nat = nat.cpu()
    with torch.no_grad():
        o_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]))
        want = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=o_emb.repeat(G, 1), **mm).numpy()
        p_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), rounding="operands")
        want_op = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=p_emb.repeat(G, 1), rounding="operands", **mm).numpy()
        # the reference's own precision: every op rounded to bfloat16 (its compute dtype)
        b_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), emulate_bf16=True)
        ref16 = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=b_emb.repeat(G, 1), emulate_bf16=True, **mm).numpy()
    scale = o_emb.abs().max().item()
    ev, ev_op = (nat - o_emb).abs().max().item() / scale, (nat - p_emb).abs().max().item() / scale
    r, r_op, r16 = rel_err(lp, want), rel_err(lp, want_op), rel_err(ref16, want)
    ev16 = (b_emb - o_emb).abs().max().item() / scale
    print("7B, 28 layers / 32 ViT blocks: vision rel err vs fp32 %.2e, vs the operand-rounded oracle %.2e" % (ev, ev_op))
    print("7B, 28 layers: logp rel err vs fp32 mean %.2e p95 %.2e max %.2e | vs the operand-rounded oracle mean %.2e p95 %.2e max %.2e"
          % (r.mean(), np.quantile(r, 0.95), r.max(), r_op.mean(), np.quantile(r_op, 0.95), r_op.max()))
    print("7B, 28 layers: the reference's own bf16 execution vs fp32: vision %.2e, logp mean %.2e max %.2e" % (ev16, r16.mean(), r16.max()))
    # At 28 layers of 3584 / 18944-long bf16 contractions no bf16 pipeline holds 1e-3 against fp32 -- and none holds it against
    # another bf16 pipeline either (the operand-rounded oracle is as far from the engine as fp32 is: rounding is discontinuous, see
    # tests/test_7b_width_oracle_gpu.py, where every layer alone is held to 1e-3 on the mean).  The bar at full depth: closer to fp32 than the
    # reference's own compute dtype, plus absolute caps.
    assert ev <= ev16 and ev < 3e-2
    assert r.mean() <= r16.mean() * 1.1 and r.mean() < 6e-3 and r.max() < 2e-2, (r.mean(), r.max(), r16.mean())
    assert r_op.mean() < 3 * r.mean() and ev_op < 3e-2
