"""GPU: the update half -- training forward log-probs, hand-written backward (weight gradients) and the AdamW step -- against
torch autograd / torch.optim.AdamW on the fp32 CPU oracle, same seeded weights and tokens (tiny config, GQA 2 and 7)."""
import numpy as np
import pytest
import torch

from oracle import grpo as og
from oracle import ints
from oracle import model as om
from tests.helpers import make_engine, pixels_from_frames, rel_err
from video_r2_amd.host.processor import video_prompt_ids

pytestmark = pytest.mark.gpu


def _setup(group, seed, G=3, C=10):
    ocfg = om.cfg_tiny(group)
    eng, w, cfg = make_engine(ocfg, seed)
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    prompt = np.asarray(video_prompt_ids(cfg, grid[0] * grid[1] * grid[2] // 4, [11, 12, 13], [21, 22, 23, 24, 25]))
    comp = torch.randint(10, 2000, (G, C), generator=g)
    comp[1, 6] = cfg.eos_token_id                       # one sequence ends early: masked tail
    return ocfg, eng, w, cfg, px, grid, prompt, comp


def _oracle_grads(ocfg, w, px, grid, prompt, comp, adv, mask, ref_lp, loss_type, beta=0.04):
    """torch autograd through the fp32 oracle: returns (logps, loss, {hf tensor name: grad})."""
    wt = {k: (v.clone().requires_grad_(k.startswith("model.") or k == "lm_head.weight")) for k, v in w.items()}
    G, C = comp.shape
    with torch.no_grad():
        emb = om.vision_forward(w, ocfg, torch.from_numpy(px), np.asarray([grid]))
    full = torch.cat([torch.from_numpy(np.repeat(prompt[None], G, 0)), comp], 1)
    lp = om.per_token_logps(wt, ocfg, full, np.ones_like(full.numpy()), C, 0.9, video_embeds=emb.repeat(G, 1),
                            video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    m = torch.from_numpy(mask).float()
    A = torch.from_numpy(adv).unsqueeze(1)
    kl = torch.exp(ref_lp - lp) - (ref_lp - lp) - 1
    c1 = torch.exp(lp - lp.detach())
    ptl = -torch.min(c1 * A, torch.clamp(c1, 0.8, 1.2) * A) + beta * kl
    if loss_type == "grpo":
        loss = ((ptl * m).sum(-1) / m.sum(-1).clamp(min=1)).mean()
    else:
        loss = (ptl * m).sum() / m.sum().clamp(min=1)
    loss.backward()
    return lp.detach(), loss.item(), {k: v.grad for k, v in wt.items() if v.requires_grad and v.grad is not None}


def _engine_grads_as_hf(tr, cfg):
    """TrainState gradients (fused / interleaved engine layout) -> HF tensor names."""
    from video_r2_amd.weights import deinterleave_gate_up
    t = cfg.text
    st = tr.state
    hd, nq, nkv, H, I = t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.hidden_size, t.intermediate_size
    a, c = nq * hd, (nq + nkv) * hd
    out = {"model.embed_tokens.weight": st.grad["embed"], "model.norm.weight": st.grad["norm"], "lm_head.weight": st.grad["lm_head"][:, :H]}
    for i in range(t.num_hidden_layers):
        p = f"model.layers.{i}."
        qkv, qb = st.grad[f"l{i}.qkv_w"][:, :H], st.grad[f"l{i}.qkv_b"]
        out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = qkv[:a], qkv[a:c], qkv[c:]
        out[p + "self_attn.q_proj.bias"], out[p + "self_attn.k_proj.bias"], out[p + "self_attn.v_proj.bias"] = qb[:a], qb[a:c], qb[c:]
        out[p + "self_attn.o_proj.weight"] = st.grad[f"l{i}.o_w"][:, :a]
        gg, gu = deinterleave_gate_up(st.grad[f"l{i}.gu_w"][:, :H].contiguous(), I)
        out[p + "mlp.gate_proj.weight"], out[p + "mlp.up_proj.weight"] = gg, gu
        out[p + "mlp.down_proj.weight"] = st.grad[f"l{i}.down_w"][:, :I]
        out[p + "input_layernorm.weight"] = st.grad[f"l{i}.ln1"]
        out[p + "post_attention_layernorm.weight"] = st.grad[f"l{i}.ln2"]
    return {k: v.float().cpu() for k, v in out.items()}


@pytest.mark.parametrize("group,loss_type,keep_logits", [(2, "bnpo", True), (7, "grpo", True), (2, "bnpo", False)])
def test_forward_backward_vs_autograd(group, loss_type, keep_logits):
    """keep_logits False: the head backward recomputes the logits chunk by chunk (groups whose fp32 logits exceed
    Trainer.KEEP_LOGITS_BYTES) instead of reading the ones forward() kept"""
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.train import Trainer
    ocfg, eng, w, cfg, px, grid, prompt, comp = _setup(group, 40 + group)
    tr = Trainer(eng, eng.text)
    if not keep_logits:
        tr.KEEP_LOGITS_BYTES = 0
    emb, plan = eng.encode_video(torch.from_numpy(px).to(eng.device), [grid])
    lp = tr.forward(prompt, emb, plan, [1.0], comp, 0.9)
    G, C = comp.shape
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = og.advantages(np.asarray([[1.0], [0.0], [0.5]], dtype=np.float32), [1.0], G)["advantages"]
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        ref_lp = lp.cpu() + 0.05 * torch.randn(G, C, generator=gen)            # stand-in reference-model log-probs
    out = hg.grpo_loss(lp, torch.from_numpy(mask).to(eng.device), torch.from_numpy(adv).to(eng.device), ref_logps=ref_lp.to(eng.device),
                       beta=0.04, loss_type=loss_type, stream=torch.cuda.current_stream().cuda_stream)
    tr.backward(out["dlogp"])
    eng.sync()
    o_lp, o_loss, o_grads = _oracle_grads(ocfg, w, px, grid, prompt, comp, adv, mask, ref_lp, loss_type)
    r = rel_err(lp.cpu().numpy(), o_lp.numpy())[mask.astype(bool)]
    assert np.quantile(r, 0.95) < 1e-3 and r.max() < 2e-3, (np.quantile(r, 0.95), r.max())       # bf16-operand floor
    assert abs(out["loss"].item() - o_loss) <= 1e-3 * max(abs(o_loss), 1e-2)
    got = _engine_grads_as_hf(tr, cfg)
    assert set(got) == set(o_grads)
    worst = {}
    for k, gref in o_grads.items():
        e = (got[k] - gref).norm().item() / max(gref.norm().item(), 1e-12)
        worst[k] = e
        # bf16 operands in every backward GEMM (dY and X rounded to 8 bits): 3e-2 of each tensor's gradient norm
        assert e < 3e-2, (k, e, gref.norm().item())
    print("max rel grad err", max(worst.values()), "median", float(np.median(list(worst.values()))))
    # tokens never fed to the LM get exactly zero embedding gradient; video placeholder rows too
    ge = got["model.embed_tokens.weight"]
    used = set(prompt.tolist()) | set(comp.numpy().reshape(-1).tolist())
    assert ge[cfg.video_token_id].abs().max().item() == 0
    assert all(ge[i].abs().max().item() == 0 for i in range(0, cfg.text.vocab_size, 37) if i not in used)


def test_optimizer_step_matches_torch_adamw_and_changes_the_policy():
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.train import Trainer, cosine_lr
    from video_r2_amd.weights import unpack_tiles
    ocfg, eng, w, cfg, px, grid, prompt, comp = _setup(2, 51)
    tr = Trainer(eng, eng.text)
    emb, plan = eng.encode_video(torch.from_numpy(px).to(eng.device), [grid])
    G, C = comp.shape
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = torch.tensor([1.0, -1.0, 0.3], device=eng.device)
    lp0 = tr.forward(prompt, emb, plan, [1.0], comp, 0.9).clone()
    out = hg.grpo_loss(lp0, torch.from_numpy(mask).to(eng.device), adv, ref_logps=lp0, beta=0.04, loss_type="bnpo",
                       stream=torch.cuda.current_stream().cuda_stream)
    tr.backward(out["dlogp"])
    eng.sync()
    name = "l1.o_w"
    g = tr.state.grad[name].clone(); p_before = tr.state.master[name].clone()
    total = torch.sqrt(sum((x.double() ** 2).sum() for x in tr.state.grad.values())).item()
    lr = 1e-3
    gn = tr.optimizer_step(lr=lr, weight_decay=0.01, max_grad_norm=1.0)
    eng.sync()
    assert abs(gn.item() - total) < 1e-3 * total
    coef = min(1.0, 1.0 / (total + 1e-6))
    ref = torch.nn.Parameter(p_before.cpu().clone())
    opt = torch.optim.AdamW([ref], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ref.grad = (g.cpu() * coef)
    opt.step()
    assert (tr.state.master[name].cpu() - ref.detach()).abs().max().item() < 1e-6
    n, kp = eng.text.shape[name]
    want = tr.state.master[name].to(torch.bfloat16)
    assert torch.equal(unpack_tiles(eng.text[name], n, kp), want)                              # forward copy re-emitted
    kt, npad = eng.text.shape[name + "_t"]
    assert torch.equal(unpack_tiles(eng.text[name + "_t"], kt, npad)[:, :n], want.t())          # dgrad (transposed) copy too
    assert all(float(x.abs().max()) == 0 for x in tr.state.grad.values())                                  # gradients zeroed
    # the policy moved in the direction the advantages ask for: sequence 0 (A>0) more likely, sequence 1 (A<0) less
    lp1 = tr.forward(prompt, emb, plan, [1.0], comp, 0.9)
    m = torch.from_numpy(mask).to(eng.device).float()
    d = ((lp1 - lp0) * m).sum(1)
    assert d[0].item() > 0 and d[1].item() < 0
    assert cosine_lr(1e-6, 0, 100) == 1e-6 and abs(cosine_lr(1e-6, 50, 100) - 5e-7) < 1e-12 and cosine_lr(1e-6, 100, 100) < 1e-12


@pytest.mark.parametrize("group,lens", [(2, [10, 7, 1]), (7, [3, 0, 10])])
def test_packed_completions_equal_the_padded_layout(group, lens):
    """Completions cut at their first EOS are scored and trained on as PACKED rows (Engine.score / Trainer.forward with lens=):
    sum(lens) rows instead of G x C.  A row's arithmetic does not depend on which other rows share its launch, so the log-probs
    of the kept positions equal the padded layout's (bit for bit in the GEMMs; the attention tiles start at the sequence's own
    first row either way), the skipped positions read 0, and with the loss gradient zero behind each EOS the weight gradients of
    the two layouts agree to fp32 summation order (the token dimension is the GEMMs' k)."""
    from video_r2_amd.train import Trainer
    ocfg, eng, w, cfg, px, grid, prompt, comp = _setup(group, 70 + group)
    G, C = comp.shape
    lens = np.asarray(lens)
    valid = torch.from_numpy(np.arange(C)[None, :] < lens[:, None])
    emb, plan = eng.encode_video(torch.from_numpy(px).to(eng.device), [grid])
    st = eng.prefill(eng.text, prompt, emb, plan, [1.0])
    dense = eng.score(eng.text, st, comp, 0.9).cpu().clone()
    packed = eng.score(eng.text, st, comp, 0.9, lens=lens).cpu().clone()
    assert torch.equal(packed[~valid], torch.zeros_like(packed[~valid]))
    assert (packed[valid] - dense[valid]).abs().max().item() <= 1e-5
    full = torch.cat([torch.from_numpy(np.repeat(prompt[None], G, 0)), comp], 1)
    want = om.per_token_logps(w, ocfg, full, np.ones_like(full.numpy()), C, 0.9, pixel_values_videos=torch.from_numpy(np.tile(px, (G, 1))),
                              video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    r = rel_err(packed.numpy(), want.numpy())[valid.numpy()]
    assert r.max() < 2e-3, r.max()
    assert torch.equal(eng.score(eng.text, st, comp, 0.9, lens=[C] * G).cpu(), dense)     # full lengths = the dense layout
    assert float(eng.score(eng.text, st, comp, 0.9, lens=[0] * G).abs().max()) == 0
    # training forward + backward, packed against padded with the same (masked) loss gradient
    tr = Trainer(eng, eng.text)
    gen = torch.Generator().manual_seed(3)
    dl = (torch.randn(G, C, generator=gen) * valid).to(eng.device)
    lp_d = tr.forward(prompt, emb, plan, [1.0], comp, 0.9).cpu().clone()
    tr.backward(dl)
    eng.sync()
    g_d = {k: v.clone() for k, v in tr.state.grad.items()}
    tr.state.zero_grad()
    lp_p = tr.forward(prompt, emb, plan, [1.0], comp, 0.9, lens=lens).cpu().clone()
    tr.backward(dl)
    eng.sync()
    assert torch.equal(lp_p[~valid], torch.zeros_like(lp_p[~valid]))
    assert (lp_p[valid] - lp_d[valid]).abs().max().item() <= 1e-5
    for k, gd in g_d.items():
        e = (tr.state.grad[k] - gd).norm().item() / max(gd.norm().item(), 1e-20)
        assert e < 2e-3, (k, e)


@pytest.mark.parametrize("group,lens", [(2, None), (2, [10, 7, 1]), (7, [3, 0, 10]), (7, None)])
def test_unified_scoring_is_bit_identical_to_the_training_forward(group, lens):
    """The two passes whose difference is the loss -- the no-grad scoring of the reference / old policy (Engine.score_unified) and
    the policy's training forward (Trainer.forward) -- run the same launches over the same row space: with identical weights every
    log-prob is identical in every bit (kl == 0 and ratio == 1 exactly, as in the reference where both come from one model call,
    grpo_trainer.py:723-741 at :826 and :1094), for dense and for packed completions, and from run to run."""
    from video_r2_amd.train import Trainer
    ocfg, eng, w, cfg, px, grid, prompt, comp = _setup(group, 90 + group)
    G, C = comp.shape
    ln = None if lens is None else np.asarray(lens)
    emb, plan = eng.encode_video(torch.from_numpy(px).to(eng.device), [grid])
    emb = emb.clone()
    tr = Trainer(eng, eng.text)
    a = eng.score_unified(eng.text, prompt, emb, plan, [1.0], comp, 0.9, tag="ref", lens=ln).clone()
    b = tr.forward(prompt, emb, plan, [1.0], comp, 0.9, lens=ln).clone()
    a2 = eng.score_unified(eng.text, prompt, emb, plan, [1.0], comp, 0.9, tag="ref", lens=ln).clone()
    eng.sync()
    assert torch.equal(a, b) and torch.equal(a, a2)
    # and it is the same function as the prefix-cache scoring pass, to fp32 summation order
    st = eng.prefill(eng.text, prompt, emb, plan, [1.0])
    c = eng.score(eng.text, st, comp, 0.9, lens=ln)
    eng.sync()
    assert (a - c).abs().max().item() <= 2e-3 and bool((a != 0).any())
    if ln is not None:
        valid = torch.from_numpy(np.arange(C)[None, :] < ln[:, None]).to(a.device)
        assert not a[~valid].any()
