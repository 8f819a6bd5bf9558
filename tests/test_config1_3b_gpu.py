"""GPU: BASELINE.json configs[0] as a parity case -- Qwen2.5-VL-3B shapes (36 layers, 16 q / 2 kv heads, tied embeddings,
vocab 151936), 4 frames of 224^2, G = 2, one prompt -- engine rollout-scoring path vs the fp32 CPU oracle on the same seeded
random-init weights.  Slow (~2 min: the oracle runs the 3.75 B-parameter model on the host cores)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg

pytestmark = pytest.mark.gpu


def test_3b_4frames_g2_logps_vs_oracle():
    from video_r2_amd import weights as W
    from video_r2_amd.engine import Engine
    from video_r2_amd.host.processor import video_prompt_ids
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ocfg = om.cfg_3b()
    w = om.make_weights(ocfg, seed=3)                                   # bf16-representable fp32 values, ~15 GB on the host
    cfg = to_engine_cfg(ocfg)
    dev = "cuda:0"
    vis = W.build_vision(cfg, w, dev)
    txt = W.build_text(cfg, w, dev)
    eng = Engine(cfg, vis, txt, device=dev)
    g = torch.Generator().manual_seed(7)
    frames = torch.randint(0, 256, (4, 3, 224, 224), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    assert grid == (2, 16, 16) and px.shape == (512, 1176)
    txt_ids = torch.randint(1000, 150000, (62,), generator=g).tolist()
    prompt = np.asarray(video_prompt_ids(cfg, 128, txt_ids[:8], txt_ids[8:]))      # S = 128 + 64 = 192
    assert prompt.shape[0] == 192
    G, C = 2, 12
    emb, plan = eng.encode_video(torch.from_numpy(px).to(dev), [grid])
    st = eng.prefill(txt, prompt, emb, plan, [1.0])
    toks, _ = eng.decode(txt, st, G, C, temperature=0.9, top_k=50, seed=11)
    toks = torch.clamp(toks, max=cfg.image_token_id - 1)
    lp = eng.score(txt, st, toks, 0.9)
    eng.sync()
    comp = toks.cpu().numpy().astype(np.int64)
    full = np.concatenate([np.repeat(prompt[None], G, 0), comp], 1)
    mm = dict(video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    ids, am = torch.from_numpy(full), np.ones_like(full)
    with torch.no_grad():
        o_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]))
        want = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=o_emb.repeat(G, 1), **mm).numpy()
        # the reference's own precision: the same restatement with every op rounded to bfloat16 (its compute dtype)
        b_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), emulate_bf16=True)
        ref16 = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=b_emb.repeat(G, 1), emulate_bf16=True, **mm).numpy()
        # the engine's OWN rounding points inside the oracle (oracle/model.py _rounders "operands")
        p_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), rounding="operands")
        want_op = om.per_token_logps(w, ocfg, ids, am, C, 0.9, video_embeds=p_emb.repeat(G, 1), rounding="operands", **mm).numpy()
    nat = torch.empty_like(emb)
    nat[torch.from_numpy(plan.window_index).to(emb.device)] = emb
    scale = o_emb.abs().max().item()
    ev = (nat.cpu() - o_emb).abs().max().item() / scale
    ev16 = (b_emb - o_emb).abs().max().item() / scale
    r = rel_err(lp.cpu().numpy(), want)
    r16 = rel_err(ref16, want)
    r_op = rel_err(lp.cpu().numpy(), want_op)
    ev_op = (nat.cpu() - p_emb).abs().max().item() / p_emb.abs().max().item()
    print("3B config-1 vs the operand-rounded oracle: vision rel err %.2e; logp rel err mean %.2e p95 %.2e max %.2e"
          % (ev_op, r_op.mean(), np.quantile(r_op, 0.95), r_op.max()))
    # reported, not a bar: rounding where the engine rounds does not bring the oracle closer than fp32 is (bf16 rounding is
    # discontinuous: see tests/test_7b_width_oracle_gpu.py, where every layer ALONE is held to 1e-3 on the mean at the 7B widths)
    assert r_op.mean() < 3 * max(r.mean(), 1e-4) and ev_op < 3e-2
    print("3B config-1: vision rel err ours", ev, "bf16-reference", ev16)
    print("3B config-1: logp rel err ours mean %.2e p95 %.2e max %.2e | bf16-reference mean %.2e p95 %.2e max %.2e"
          % (r.mean(), np.quantile(r, 0.95), r.max(), r16.mean(), np.quantile(r16, 0.95), r16.max()))
    # At 36 layers bf16-operand noise alone exceeds the 1e-3 of the small golden cases; the bar here is the
    # reference's own deviation from fp32 when run in its compute dtype, plus an absolute cap.
    assert ev <= ev16 and ev < 3e-2
    assert r.mean() <= r16.mean() and np.quantile(r, 0.95) <= np.quantile(r16, 0.95)
    assert np.quantile(r, 0.95) < 5e-3 and r.max() < 1e-2
