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
    # FINDING (round 4, measured here and at 28 / 36 layers): rounding where the engine rounds does NOT bring the oracle closer to
    # the engine than fp32 is -- bf16 rounding is discontinuous, two pipelines that differ by d << ulp before a rounding point differ
    # by ~sqrt(d * ulp) after it, so within a few rounding points any two bf16-operand pipelines that are not bit-identical sit one
    # rounding error apart, exactly as far as each sits from fp32.  Depth-compounded log-probs therefore cannot certify 1e-3 for ANY
    # bf16 pipeline (the reference's own bf16 execution included, r16 below); what CAN be held to 1e-3 (on the mean) at these widths is every
    # layer on its own, from identical inputs: test_7b_widths_every_layer_alone below (every position of layers 1-3 within 1e-3).
    assert r_op.mean() < 3 * max(r.mean(), 1e-4) and ev_op < 1e-2
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


def test_7b_widths_every_layer_alone():
    """north_star's 1e-3 at the REAL widths (hidden 3584, 28 q / 4 kv heads of 128, MLP 18944, vocabulary 152064), one
    decoder layer at a time from identical inputs: layer l of a 4-layer model receives the residual stream the oracle computed for
    it (fp32 rows scattered into the prompt as vision features), runs through the engine's prefill kernels -- RMSNorm, q/k/v GEMM,
    M-RoPE, causal GQA attention, o GEMM, SwiGLU GEMM, down GEMM, final norm, lm_head -- and its log-probs over the full vocabulary
    are compared with the oracle's for the same layer, in fp32 and with the engine's rounding points.  No compounding over depth:
    what is left is the arithmetic of one layer + head at 3584 / 18944 / 152064-long contractions.
    Measured (round 4): layers 1-3, whose input is a residual stream of ordinary size: max 6.1-7.6e-4 against the operand-rounded
    oracle, 8.1-9.0e-4 against fp32 (mean 1.6-2.3e-4) -- EVERY position within north_star's 1e-3.  Layer 0, fed raw embedding rows
    (norm ~1, an order of magnitude below what the layers add to it, so the branch outputs' rounding noise is the whole signal):
    mean 4.4e-4 / max 2.0e-3 (fp32: 6.8e-4 / 2.6e-3) -- the mean holds, single positions do not.  Seven bf16 rounding points sit
    between a layer's input and the logits; the operand-rounded oracle is only 1.2-1.5x closer than fp32, because a rounding point
    turns a difference d << ulp between two pipelines into ~sqrt(d ulp): after a few of them any two bf16-operand pipelines that are
    not bit-identical are most of a rounding error apart.  Bit-level agreement on identical operands is what the kernel tests hold
    (exact-integer GEMM / GEMV / attention cases in tests/test_kernels_gpu.py)."""
    from video_r2_amd import weights as W
    from video_r2_amd.engine import Engine
    torch.set_num_threads(min(32, torch.get_num_threads()))
    full = om.cfg_7b()
    L = 4
    ocfg = om.Cfg(vision=om.VisionCfg(**{**full.vision.__dict__, "depth": 1, "fullatt_block_indexes": (0,)}),
                  text=om.TextCfg(**{**full.text.__dict__, "num_hidden_layers": L}))
    w = om.make_weights(ocfg, seed=7)
    dev = "cuda:0"
    grid = (8, 16, 16)
    N = 512
    ids = np.asarray([ocfg.vision_start_token_id] + [ocfg.video_token_id] * N + [ocfg.vision_end_token_id] + [1234, 5678])
    S = len(ids)
    g = torch.Generator().manual_seed(11)
    # the residual stream entering every layer of a 4-layer run over random token embeddings (operand-rounded oracle)
    tok = torch.randint(1000, 150000, (1, S), generator=g)
    from oracle import vision_index as ovi
    pos, _ = ovi.rope_index(ids[None], None, np.asarray([grid]), [1.0], video_token_id=ocfg.video_token_id,
                            image_token_id=ocfg.image_token_id, merge_size=2, tokens_per_second=2)
    xs = []
    with torch.no_grad():
        om.lm_forward(w, ocfg, w["model.embed_tokens.weight"][tok], torch.from_numpy(pos), rounding="operands", layer_inputs=xs)
    tgt = torch.randint(0, ocfg.text.vocab_size, (S,), generator=g)
    vcfg = to_engine_cfg(ocfg)
    vis = W.build_vision(vcfg, w, dev)
    worst = 0.0
    for l in range(L):
        one = om.Cfg(vision=ocfg.vision, text=om.TextCfg(**{**ocfg.text.__dict__, "num_hidden_layers": 1}))
        w1 = {k: v for k, v in w.items() if not k.startswith("model.layers.")}
        w1.update({k.replace(f"model.layers.{l}.", "model.layers.0."): v for k, v in w.items() if k.startswith(f"model.layers.{l}.")})
        X = xs[l][0, 1:1 + N].contiguous()                                   # fp32 rows for the video positions
        with torch.no_grad():
            lo32 = om.forward_logits(w1, one, torch.from_numpy(ids)[None], None, None, np.asarray([grid]), [1.0], video_embeds=X)[0]
            loop = om.forward_logits(w1, one, torch.from_numpy(ids)[None], None, None, np.asarray([grid]), [1.0], video_embeds=X, rounding="operands")[0]
        cfg1 = to_engine_cfg(one)
        txt1 = W.build_text(cfg1, w1, dev)
        eng = Engine(cfg1, vis, txt1, device=dev)
        plan = eng.vision_plan([grid])
        emb_w = X[torch.from_numpy(plan.window_index)].to(dev)                # row i of the features is natural token window_index[i]
        lg = torch.empty((S, cfg1.text.vocab_size), dtype=torch.float32, device=dev)
        eng.prefill(txt1, ids, emb_w, plan, [1.0], tag="fwd", all_logits=lg)
        eng.sync()
        pick = lambda z: torch.log_softmax(z.float() / 0.9, -1).gather(1, tgt[:, None])[:, 0].numpy()
        got, want32, want_op = pick(lg.cpu()), pick(lo32), pick(loop)
        r32, rop = rel_err(got, want32), rel_err(got, want_op)
        print("7B widths, layer %d alone: logp rel err vs fp32 mean %.2e max %.2e | vs the operand-rounded oracle mean %.2e max %.2e"
              % (l, r32.mean(), r32.max(), rop.mean(), rop.max()))
        worst = max(worst, rop.max())
        if l >= 1:     # north_star's tolerance on EVERY position
            assert rop.max() <= 1e-3 and r32.max() <= 1.2e-3 and r32.mean() <= 5e-4, (l, rop.max(), r32.mean(), r32.max())
        else:          # embedding-row input: the tolerance on the mean, single positions within 3e-3
            assert rop.mean() <= 1e-3 and rop.max() <= 3e-3 and r32.mean() <= 1e-3 and r32.max() <= 4e-3, (l, rop.max(), r32.max())
        del eng, txt1, lg
        torch.cuda.empty_cache()
