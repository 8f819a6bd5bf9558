"""GPU: training the vision side (--freeze_vision_tower False / --freeze_merger False; reference train_grpo.py:48-57).
VisionTrainer.forward reproduces Engine.encode_videos bit for bit while keeping the activations; VisionTrainer.backward (merger:
fc2, erf-GELU, fc1, ln_q; tower: SwiGLU MLP with biases, RMSNorm, rotary, non-causal windowed / per-frame attention through the
head_dim-128 backward on zero-padded 80-wide heads, patch embedding) against torch autograd through the fp32 CPU oracle, tensor by
tensor, together with the LM's gradients of the same step; the optimiser step moves the tower and the next encode sees it."""
import numpy as np
import pytest
import torch

from oracle import grpo as og
from oracle import ints
from oracle import model as om
from tests.helpers import make_engine, pixels_from_frames, rel_err
from tests.test_train_gpu import _engine_grads_as_hf
from video_r2_amd.host.processor import video_prompt_ids

pytestmark = pytest.mark.gpu


def _vision_grads_as_hf(vs, cfg):
    """VisionState gradients (fused / interleaved / K-padded engine layout) -> HF tensor names"""
    from video_r2_amd.weights import deinterleave_gate_up
    v = cfg.vision
    Hv, I = v.hidden_size, v.intermediate_size
    g, out = vs.grad, {}
    if "patch_w" in g:
        out["visual.patch_embed.proj.weight"] = g["patch_w"][:, :v.patch_dim].reshape(Hv, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
        for i in range(v.depth):
            s = f"visual.blocks.{i}."
            out[s + "norm1.weight"], out[s + "norm2.weight"] = g[f"b{i}.norm1"], g[f"b{i}.norm2"]
            out[s + "attn.qkv.weight"], out[s + "attn.qkv.bias"] = g[f"b{i}.qkv_w"][:, :Hv], g[f"b{i}.qkv_b"]
            out[s + "attn.proj.weight"], out[s + "attn.proj.bias"] = g[f"b{i}.proj_w"][:, :Hv], g[f"b{i}.proj_b"]
            gg, gu = deinterleave_gate_up(g[f"b{i}.gu_w"][:, :Hv].contiguous(), I)
            out[s + "mlp.gate_proj.weight"], out[s + "mlp.up_proj.weight"] = gg, gu
            bg, bu = deinterleave_gate_up(g[f"b{i}.gu_b"][:, None].contiguous(), I)
            out[s + "mlp.gate_proj.bias"], out[s + "mlp.up_proj.bias"] = bg[:, 0], bu[:, 0]
            out[s + "mlp.down_proj.weight"], out[s + "mlp.down_proj.bias"] = g[f"b{i}.down_w"][:, :I], g[f"b{i}.down_b"]
    if "fc1_w" in g:
        mh = Hv * v.spatial_merge_size ** 2
        out["visual.merger.ln_q.weight"] = g["ln_q"]
        out["visual.merger.mlp.0.weight"], out["visual.merger.mlp.0.bias"] = g["fc1_w"][:, :mh], g["fc1_b"]
        out["visual.merger.mlp.2.weight"], out["visual.merger.mlp.2.bias"] = g["fc2_w"][:, :mh], g["fc2_b"]
    return {k: t.float().cpu() for k, t in out.items()}


def _setup(group, seed, frames_hw=(4, 56, 84), G=3, C=10):
    ocfg = om.cfg_tiny(group)
    eng, w, cfg = make_engine(ocfg, seed)
    g = torch.Generator().manual_seed(seed)
    n, h, ww = frames_hw
    frames = torch.randint(0, 256, (n, 3, h, ww), generator=g, dtype=torch.uint8)
    px, grid = pixels_from_frames(frames.numpy(), cfg.vision)
    prompt = np.asarray(video_prompt_ids(cfg, grid[0] * grid[1] * grid[2] // 4, [11, 12, 13], [21, 22, 23, 24, 25]))
    comp = torch.randint(10, 2000, (G, C), generator=g)
    comp[1, 6] = cfg.eos_token_id
    return ocfg, eng, w, cfg, frames, px, grid, prompt, comp


def _oracle_grads_with_vision(ocfg, w, px, grid, prompt, comp, adv, mask, ref_lp, train):
    """torch autograd through the fp32 oracle, vision tower included: (logps, {hf name: grad}) for LM + the trainable vision tensors"""
    def trainable(k):
        if k.startswith("model.") or k == "lm_head.weight":
            return True
        if k.startswith("visual.merger."):
            return train[1]
        return k.startswith("visual.") and train[0]
    wt = {k: v.clone().requires_grad_(trainable(k)) for k, v in w.items()}
    G, C = comp.shape
    emb = om.vision_forward(wt, ocfg, torch.from_numpy(px), np.asarray([grid]))
    full = torch.cat([torch.from_numpy(np.repeat(prompt[None], G, 0)), comp], 1)
    lp = om.per_token_logps(wt, ocfg, full, np.ones_like(full.numpy()), C, 0.9, video_embeds=emb.repeat(G, 1),
                            video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
    m = torch.from_numpy(mask).float()
    A = torch.from_numpy(adv).unsqueeze(1)
    kl = torch.exp(ref_lp - lp) - (ref_lp - lp) - 1
    c1 = torch.exp(lp - lp.detach())
    ptl = -torch.min(c1 * A, torch.clamp(c1, 0.8, 1.2) * A) + 0.04 * kl
    loss = (ptl * m).sum() / m.sum().clamp(min=1)
    loss.backward()
    return lp.detach(), {k: v.grad for k, v in wt.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize("group,train", [(2, (True, True)), (2, (False, True)), (7, (True, False))])
def test_vision_forward_is_the_engines_and_backward_matches_autograd(group, train):
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.train import Trainer
    from video_r2_amd.train_vision import VisionTrainer
    ocfg, eng, w, cfg, frames, px, grid, prompt, comp = _setup(group, 60 + group)
    clips = [dict(frames_u8=frames, resize_to=None)]
    (emb0, plan), = eng.encode_videos(clips, per_prompt=[1])
    emb0 = emb0.clone()
    vt = VisionTrainer(eng, eng.vision, train_tower=train[0], train_merger=train[1])
    tr = Trainer(eng, eng.text, vision_trainer=vt)
    rows, plan2 = eng.encode_videos(clips, per_prompt=[1], rows_only=True)
    assert plan2 is plan
    lp = tr.forward(prompt, emb0, plan, [1.0], comp, 0.9, vision_rows=rows)
    eng.sync()
    assert torch.equal(eng.buf("tv.out", emb0.shape, torch.float32), emb0)        # the training forward's tower == the no-grad tower, bit for bit
    G, C = comp.shape
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = og.advantages(np.asarray([[1.0], [0.0], [0.5]], dtype=np.float32), [1.0], G)["advantages"]
    with torch.no_grad():
        ref_lp = lp.cpu() + 0.05 * torch.randn(G, C, generator=torch.Generator().manual_seed(1))
    out = hg.grpo_loss(lp, torch.from_numpy(mask).to(eng.device), torch.from_numpy(adv).to(eng.device), ref_logps=ref_lp.to(eng.device),
                       beta=0.04, loss_type="bnpo", stream=torch.cuda.current_stream().cuda_stream)
    tr.backward(out["dlogp"])
    eng.sync()
    o_lp, o_grads = _oracle_grads_with_vision(ocfg, w, px, grid, prompt, comp, adv, mask, ref_lp, train)
    r = rel_err(lp.cpu().numpy(), o_lp.numpy())[mask.astype(bool)]
    assert np.quantile(r, 0.95) < 1e-3 and r.max() < 2e-3
    got = dict(_engine_grads_as_hf(tr, cfg))
    got.update(_vision_grads_as_hf(vt.state, cfg))
    assert set(got) == set(o_grads), sorted(set(got) ^ set(o_grads))[:6]
    worst = {}
    for k, gref in o_grads.items():
        worst[k] = (got[k] - gref).norm().item() / max(gref.norm().item(), 1e-12)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("vision train %s: max rel grad err %.3e (median %.3e); worst: %s"
          % (train, max(worst.values()), float(np.median(list(worst.values()))), ", ".join("%s %.2e" % kv for kv in top)))
    for k, e in worst.items():                 # bf16 operands in every backward GEMM: 3e-2 of each tensor's gradient norm, as for the LM
        assert e < 3e-2, (k, e)
    n_vis = sum(k.startswith("visual.") for k in o_grads)
    assert n_vis == (5 if train[1] else 0) + ((1 + 12 * cfg.vision.depth) if train[0] else 0)


def test_optimizer_step_moves_the_tower_and_the_next_encode_sees_it():
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.train import Trainer
    from video_r2_amd.train_vision import VisionTrainer
    from video_r2_amd.weights import clone_params
    ocfg, eng, w, cfg, frames, px, grid, prompt, comp = _setup(2, 71)
    clips = [dict(frames_u8=frames, resize_to=None)]
    frozen = clone_params(eng.vision)                                  # what the KL reference keeps
    (emb0, plan), = eng.encode_videos(clips, per_prompt=[1])
    emb0 = emb0.clone()
    vt = VisionTrainer(eng, eng.vision, True, True)
    tr = Trainer(eng, eng.text, vision_trainer=vt)
    rows, _ = eng.encode_videos(clips, per_prompt=[1], rows_only=True)
    lp = tr.forward(prompt, emb0, plan, [1.0], comp, 0.9, vision_rows=rows)
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = torch.tensor([1.0, -1.0, 0.3], device=eng.device)
    out = hg.grpo_loss(lp, torch.from_numpy(mask).to(eng.device), adv, ref_logps=lp.clone(), beta=0.04, loss_type="bnpo",
                       stream=torch.cuda.current_stream().cuda_stream)
    tr.backward(out["dlogp"])
    eng.sync()
    total = torch.sqrt(sum((x.double() ** 2).sum() for st in (tr.state, vt.state) for x in st.grad.values())).item()
    g_fc2 = vt.state.grad["fc2_w"].clone(); p_fc2 = vt.state.master["fc2_w"].clone()
    gn = tr.optimizer_step(lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    eng.sync()
    assert abs(gn.item() - total) < 1e-3 * total                       # ONE global norm over the LM and the vision tensors
    # first AdamW step: p -= lr * sign-like update g / (|g| + eps) (bias-corrected moments of one step)
    coef = min(1.0, 1.0 / (total + 1e-6))
    gs = g_fc2 * coef
    want = p_fc2 - 1e-3 * gs / (gs.abs() + 1e-8)
    assert (vt.state.master["fc2_w"] - want).abs().max().item() < 1e-6
    assert float(vt.state.arena.abs().max()) == 0.0                    # gradients zeroed for the next step
    (emb1, _), = eng.encode_videos(clips, per_prompt=[1])
    (embf, _), = eng.encode_videos(clips, per_prompt=[1], vision=frozen)
    assert not torch.equal(emb1, emb0) and torch.equal(embf.clone(), emb0)   # the policy's tower moved, the frozen copy did not


def test_grpo_trainer_with_the_vision_side_trainable(tmp_path):
    """QwenGRPOTrainer with --freeze_vision_tower False --freeze_merger False (as train_grpo.py's check_trainable_set hands them over):
    step 1 still has kl == 0 exactly (the frozen copy of the tower + LM == the policy), the update moves the policy's tower and not
    the reference's, step 2 sees kl > 0, and a checkpoint carries the trained tower (HF-layout weights + optimiser state)."""
    import os
    from tests.test_model_gpu import _model
    from video_r2_amd import params, rewards
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 31)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "10", "--beta", "0.04", "--loss_type", "bnpo",
                              "--learning_rate", "1e-3", "--reward_func_names", "format_reward", "--reward_weights", "1.0",
                              "--freeze_vision_tower", "False", "--freeze_merger", "False"])
    args.train_vision_tower, args.train_merger = True, True

    def processing_fn(ex):
        frames = torch.randint(0, 256, (2, 3, 56, 56), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).cuda(), video_grid_thw=[grid], second_per_grid_ts=[1.0],
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if i % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(3)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    assert tr.updater.vis is not None and tr.ref_vision is not None
    steps = list(tr.iter_steps())
    fc2_0, qkv_0 = m.vision["fc2_w"].clone(), m.vision["b0.qkv_w"].clone()
    o1 = tr.step(steps[0])
    assert o1["metrics"]["kl"] == 0.0 and torch.equal(o1["rollouts"][0]["ref_logps"], o1["rollouts"][0]["logps"])
    assert not torch.equal(m.vision["fc2_w"], fc2_0) and not torch.equal(m.vision["b0.qkv_w"], qkv_0)      # the policy's tower moved
    assert torch.equal(tr.ref_vision["fc2_w"], fc2_0) and torch.equal(tr.ref_vision["b0.qkv_w"], qkv_0)    # the reference's did not
    o2 = tr.step(steps[1])
    assert o2["metrics"]["kl"] > 0.0 and np.isfinite(o2["loss"])
    d = tr.save_checkpoint(str(tmp_path))
    m2 = VideoR2Model.from_pretrained(d)
    assert torch.equal(m2.vision["fc2_w"], m.vision["fc2_w"]) and torch.equal(m2.vision["b0.qkv_w"], m.vision["b0.qkv_w"])
    tr2 = QwenGRPOTrainer(m2, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names).load_checkpoint(d)
    for k in tr.updater.vis.state.master:
        assert torch.equal(tr.updater.vis.state.master[k], tr2.updater.vis.state.master[k]), k
        assert torch.equal(tr.updater.vis.state.m[k], tr2.updater.vis.state.m[k]), k
    assert os.path.exists(os.path.join(d, "trainer_state.pt"))


def test_frozen_lm_under_a_trainable_vision_side():
    """--freeze_llm True with --freeze_vision_tower False / --freeze_merger False (configure_llm + configure_vision_tower of the
    reference): the LM's backward computes input gradients only.  The vision gradients are those of the run that also trains the LM
    (same launches on the way down), the LM holds no masters / gradients / moments, its weights do not move, and
    the clip norm is the vision tensors' alone."""
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.train import Trainer
    from video_r2_amd.train_vision import VisionTrainer
    ocfg, eng, w, cfg, frames, px, grid, prompt, comp = _setup(2, 83)
    clips = [dict(frames_u8=frames, resize_to=None)]
    (emb0, plan), = eng.encode_videos(clips, per_prompt=[1])
    emb0 = emb0.clone()
    rows, _ = eng.encode_videos(clips, per_prompt=[1], rows_only=True)
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = torch.tensor([1.0, -1.0, 0.3], device=eng.device)
    grads = {}
    for mode in ("full", "frozen_lm"):
        vt = VisionTrainer(eng, eng.vision, True, True)
        tr = Trainer(eng, eng.text, vision_trainer=vt, train_lm=(mode == "full"))
        lp = tr.forward(prompt, emb0, plan, [1.0], comp, 0.9, vision_rows=rows)
        out = hg.grpo_loss(lp, torch.from_numpy(mask).to(eng.device), adv, ref_logps=lp.clone() - 0.01, beta=0.04, loss_type="bnpo",
                           stream=torch.cuda.current_stream().cuda_stream)
        tr.backward(out["dlogp"])
        eng.sync()
        grads[mode] = {k: v.clone() for k, v in vt.state.grad.items()}
        if mode == "frozen_lm":
            assert tr.state.master == {} and tr.state.arena.numel() == 0
            before = {k: eng.text[k].clone() for k in ("l0.qkv_w", "l1.down_w", "lm_head", "embed", "norm")}
            fc2 = eng.vision["fc2_w"].clone()
            gn = tr.optimizer_step(lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
            eng.sync()
            total = torch.sqrt(sum((x.double() ** 2).sum() for x in grads[mode].values())).item()
            assert abs(gn.item() - total) < 1e-3 * total
            assert all(torch.equal(eng.text[k], v) for k, v in before.items()) and not torch.equal(eng.vision["fc2_w"], fc2)
        del tr, vt
    assert set(grads["full"]) == set(grads["frozen_lm"])
    for k in grads["full"]:     # same launches on the way down; the fp32 atomic adds of dK / dV, the column sums and the norm weights
        a, b_ = grads["full"][k], grads["frozen_lm"][k]     # arrive in a different order from run to run: 1e-5 of the tensor's norm
        assert (a - b_).norm().item() <= 1e-5 * a.norm().item(), k
    assert max(float(v.abs().max()) for v in grads["full"].values()) > 0
