"""GPU: real Qwen2.5-VL-7B WIDTHS against the fp32 CPU oracle -- hidden 3584, 28 q / 4 kv heads, MLP 18944, vocab 152064,
ViT 1280 / 3420 / 5120 -> 3584, 16 frames of 224^2 (S = 576, BASELINE config 2's prompt), G = 4 -- at TRUNCATED DEPTH (4 LM
layers, 4 ViT blocks of which 2 are full-attention) so that the oracle finishes on the host cores.  test_fullsize_7b_gpu.py
checks the full-depth model through cross-path properties only; this case puts the real tile tails (3420, 18944 = 148 x 128,
152064), the GQA group of 7 and the 152k-wide log-softmax against the oracle itself.  Slow (~1-2 min)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg

pytestmark = pytest.mark.gpu


def test_7b_widths_4_layers_logps_and_multi_group_decode_vs_oracle():
    from video_r2_amd import weights as W
    from video_r2_amd.engine import Engine
    from video_r2_amd.host.processor import video_prompt_ids
    torch.set_num_threads(min(32, torch.get_num_threads()))
    full = om.cfg_7b()
    ocfg = om.Cfg(vision=om.VisionCfg(**{**full.vision.__dict__, "depth": 4, "fullatt_block_indexes": (1, 3)}),
                  text=om.TextCfg(**{**full.text.__dict__, "num_hidden_layers": 4}))
    w = om.make_weights(ocfg, seed=5)                                   # bf16-representable fp32 values, ~9 GB on the host
    cfg = to_engine_cfg(ocfg)
    dev = "cuda:0"
    vis = W.build_vision(cfg, w, dev)
    txt = W.build_text(cfg, w, dev)
    eng = Engine(cfg, vis, txt, device=dev)
    g = torch.Generator().manual_seed(9)
    frames = torch.randint(0, 256, (16, 3, 224, 224), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    assert grid == (8, 16, 16) and px.shape == (2048, 1176)
    txt_ids = torch.randint(1000, 150000, (62,), generator=g).tolist()
    prompt = np.asarray(video_prompt_ids(cfg, 512, txt_ids[:8], txt_ids[8:]))      # S = 512 + 64
    assert prompt.shape[0] == 576
    G, C = 4, 10
    emb, plan = eng.encode_video(torch.from_numpy(px).to(dev), [grid])
    emb = emb.clone()
    st = eng.prefill(txt, prompt, emb, plan, [1.0])
    toks, _ = eng.decode(txt, st, G, C, temperature=0.9, top_k=50, seed=13)
    toks = torch.clamp(toks, max=cfg.image_token_id - 1)
    lp = eng.score(txt, st, toks, 0.9)
    eng.sync()
    comp = toks.cpu().numpy().astype(np.int64)
    rows = np.concatenate([np.repeat(prompt[None], G, 0), comp], 1)
    mm = dict(video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    with torch.no_grad():
        o_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]))
        want = om.per_token_logps(w, ocfg, torch.from_numpy(rows), np.ones_like(rows), C, 0.9, video_embeds=o_emb.repeat(G, 1), **mm).numpy()
        # the reference's own precision: the same restatement with every op rounded to bfloat16 (its compute dtype)
        b_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), emulate_bf16=True)
        ref16 = om.per_token_logps(w, ocfg, torch.from_numpy(rows), np.ones_like(rows), C, 0.9, video_embeds=b_emb.repeat(G, 1),
                                   emulate_bf16=True, **mm).numpy()
        # the engine's OWN rounding points inside the oracle (bf16 matrix-multiply inputs, fp32 everything else: oracle/model.py
        # _rounders "operands"): what is left between the two is everything that is NOT an operand rounding
        p_emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]), rounding="operands")
        want_op = om.per_token_logps(w, ocfg, torch.from_numpy(rows), np.ones_like(rows), C, 0.9, video_embeds=p_emb.repeat(G, 1),
                                     rounding="operands", **mm).numpy()
    nat = torch.empty_like(emb)
    nat[torch.from_numpy(plan.window_index).to(emb.device)] = emb
    ev = (nat.cpu() - o_emb).abs().max().item() / o_emb.abs().max().item()
    ev16 = (b_emb - o_emb).abs().max().item() / o_emb.abs().max().item()
    r = rel_err(lp.cpu().numpy(), want)
    r16 = rel_err(ref16, want)
    r_op = rel_err(lp.cpu().numpy(), want_op)
    ev_op = (nat.cpu() - p_emb).abs().max().item() / p_emb.abs().max().item()
    print("7B widths, 4 layers vs the operand-rounded oracle: vision rel err %.2e; logp rel err mean %.2e p95 %.2e max %.2e"
          % (ev_op, r_op.mean(), np.quantile(r_op, 0.95), r_op.max()))
    # north_star's bar, asserted at the real widths: against an oracle that rounds where the engine rounds, 1e-3 holds on EVERY
    # position (|d| / max(|ref|, 1e-2)); the fp32 comparison below keeps the looser caps the bf16 operand floor allows
    assert r_op.max() <= 1e-3, (r_op.mean(), r_op.max())
    assert ev_op < 2e-3, ev_op
    print("7B widths, 4 layers: vision rel err ours %.2e bf16-reference %.2e; logp rel err ours mean %.2e p95 %.2e max %.2e | "
          "bf16-reference mean %.2e p95 %.2e max %.2e" % (ev, ev16, r.mean(), np.quantile(r, 0.95), r.max(), r16.mean(),
                                                           np.quantile(r16, 0.95), r16.max()))
    # At these widths (3584- and 18944-long bf16 contractions, a 152k-wide log-softmax on |logp| ~ 12) the bf16-operand noise
    # is above the 1e-3 the small golden cases hold (measured mean 1.0e-3 / p95 2.8e-3 / max 3.7e-3 here); the bar is
    # the reference's own deviation from fp32 when run in its compute dtype, plus absolute caps -- as for the 36-layer 3B case.
    assert ev <= ev16 and ev < 1e-2
    assert r.mean() <= r16.mean() and np.quantile(r, 0.95) <= np.quantile(r16, 0.95)
    assert np.quantile(r, 0.95) < 5e-3 and r.max() < 1e-2, (np.quantile(r, 0.95), r.max())
    # the 128-row decode path (16 groups x G... here 3 groups x 4 = 12 rows -> the M <= 16 kernel, then 5 groups = 20 rows -> the
    # one-workgroup-per-CU ring kernel) must reproduce the same distributions: teacher-force the tokens through a multi-group
    # decode and compare the log-prob of every forced token with the oracle's
    for n_groups in (3, 5):
        slots = eng.prefix_slots(n_groups, len(prompt), tag="pol")
        states = [eng.prefill(txt, prompt, emb, plan, [1.0], tag="pol", prefix_out=slots[p]) for p in range(n_groups)]
        forced = np.tile(toks.cpu().numpy(), (n_groups, 1))
        _, b = eng.decode(txt, states, G, C, temperature=0.9, top_k=50, seed=3, forced_tokens=forced, use_graph=False)
        eng.sync()
        last = torch.log_softmax(b["logits"].float() / 0.9, -1)           # logits of the LAST decode step: they predict token C-1
        got = last.gather(1, torch.from_numpy(forced[:, C - 1:C]).to(dev).long())[:, 0].cpu().numpy()
        rr = rel_err(got, np.tile(want[:, C - 1], n_groups))
        assert rr.max() < 1e-2 and rr.mean() < 4e-3, (n_groups, rr.max(), rr.mean())
