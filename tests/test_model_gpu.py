"""GPU: the drop-in façade (generate / per_token_logps on left-padded batches, HF-layout save/load) and one trainer step."""
import os

import numpy as np
import pytest
import torch

from oracle import ints
from oracle import model as om
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg

pytestmark = pytest.mark.gpu


def _model(group=2, seed=11):
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(group)
    w = om.make_weights(ocfg, seed=seed)
    return VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0"), w, ocfg


def test_per_token_logps_left_padded_golden(golden_dir):
    f = np.load(os.path.join(golden_dir, "fwd_tiny_g2_pad.npz"))
    m, w, ocfg = _model(2, int(f["seed"]))
    px, grid = pixels_from_frames(f["frames_u8"], m.cfg.vision)
    lp = m.per_token_logps(torch.from_numpy(f["input_ids"]), torch.from_numpy(f["attention_mask"]), int(f["C"]),
                           temperature=float(f["temperature"]), pixel_values_videos=torch.from_numpy(px).cuda(),
                           video_grid_thw=torch.tensor([grid]), second_per_grid_ts=[float(f["sec"])])
    r = rel_err(lp.cpu().numpy(), f["hf_logps"])
    assert r.max() < 1e-3, r.max()         # north_star tolerance, |d| / max(|ref|, 1e-2)


def test_generate_batch_of_groups_and_scoring_vs_oracle():
    from video_r2_amd.model import GenerationConfig
    m, w, ocfg = _model(2, 17)
    cfg = m.cfg
    g = torch.Generator().manual_seed(1)
    # two different prompts (different videos and lengths), each repeated G=3 times, left-padded to a common length
    prompts, pxs, grids = [], [], []
    for hw, ntxt in (((56, 84), 5), ((84, 56), 9)):
        frames = torch.randint(0, 256, (2, 3, *hw), generator=g, dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [21, 22, cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id] + list(range(30, 30 + ntxt))
        prompts.append(ids); pxs.append(px); grids.append(grid)
    G = 3
    S = max(len(p) for p in prompts)
    rows, masks, pv, gt = [], [], [], []
    for p, px, grid in zip(prompts, pxs, grids):
        for _ in range(G):
            rows.append([cfg.pad_token_id] * (S - len(p)) + p); masks.append([0] * (S - len(p)) + [1] * len(p))
            pv.append(px); gt.append(grid)
    ids = torch.tensor(rows); am = torch.tensor(masks)
    pvt = torch.from_numpy(np.concatenate(pv)).cuda(); gtt = torch.tensor(gt)
    C = 10
    gc = GenerationConfig(max_new_tokens=C, temperature=0.9, top_k=50, eos_token_id=-1, seed=5)
    out = m.generate(input_ids=ids, attention_mask=am, pixel_values_videos=pvt, video_grid_thw=gtt,
                     second_per_grid_ts=[1.0] * (2 * G), generation_config=gc)
    assert out.shape == (2 * G, S + C) and torch.equal(out[:, :S].cpu(), ids)
    comp = torch.clamp(out[:, S:], max=cfg.image_token_id - 1).cpu()
    assert len({tuple(r.tolist()) for r in comp[:G]}) > 1            # rows of a group are different samples
    full = torch.cat([ids, comp], 1)
    fam = torch.cat([am, torch.ones(2 * G, C, dtype=torch.long)], 1)
    lp = m.per_token_logps(full, fam, C, temperature=0.9, pixel_values_videos=pvt, video_grid_thw=gtt, second_per_grid_ts=[1.0] * (2 * G))
    want = om.per_token_logps(w, ocfg, full, fam.numpy(), C, 0.9, pixel_values_videos=pvt.cpu(), video_grid_thw=np.asarray(gt),
                              second_per_grid_ts=[1.0] * (2 * G)).numpy()
    r = rel_err(lp.cpu().numpy(), want)
    assert np.quantile(r, 0.95) < 1e-3 and r.max() < 2e-3, (np.quantile(r, 0.95), r.max())   # bf16-operand floor, see DESIGN.md
    # mismatch between video tokens and features must raise (HF modeling_qwen2_5_vl.py:1128-1132)
    bad = ids.clone(); bad[0, -1] = cfg.video_token_id
    with pytest.raises(ValueError, match="do not match"):
        m.generate(input_ids=bad, attention_mask=am, pixel_values_videos=pvt, video_grid_thw=gtt, generation_config=gc)


def test_rows_with_equal_ids_but_different_videos_are_not_merged():
    """video placeholder ids are the same for every clip of one geometry: two samples with the same question template must
    still be encoded and scored against their OWN clips (rows are copies only if ids, grids and pixel values all agree)"""
    from video_r2_amd.model import GenerationConfig
    m, w, ocfg = _model(2, 19)
    cfg = m.cfg
    g = torch.Generator().manual_seed(2)
    pxs = []
    for _ in range(2):
        frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        pxs.append(px)
    tv = grid[0] * grid[1] * grid[2] // 4
    p = [21, cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 30, 31]
    ids = torch.tensor([p, p, p, p])                                   # rows 0,1: clip A twice; rows 2,3: clip B twice
    am = torch.ones_like(ids)
    pvt = torch.from_numpy(np.concatenate([pxs[0], pxs[0], pxs[1], pxs[1]])).cuda()
    gtt = torch.tensor([grid] * 4)
    C = 6
    comp = torch.randint(32, 200, (4, C), generator=g)
    full = torch.cat([ids, comp], 1)
    fam = torch.ones_like(full)
    lp = m.per_token_logps(full, fam, C, temperature=0.9, pixel_values_videos=pvt, video_grid_thw=gtt, second_per_grid_ts=[1.0] * 4)
    want = om.per_token_logps(w, ocfg, full, fam.numpy(), C, 0.9, pixel_values_videos=pvt.cpu(), video_grid_thw=np.asarray([grid] * 4),
                              second_per_grid_ts=[1.0] * 4).numpy()
    r = rel_err(lp.cpu().numpy(), want)
    assert np.quantile(r, 0.95) < 1e-3 and r.max() < 2e-3, (np.quantile(r, 0.95), r.max())
    # scoring rows 2,3 against clip A instead (what merging by ids alone did) is visibly different
    wrong = om.per_token_logps(w, ocfg, full[2:], fam[2:].numpy(), C, 0.9, pixel_values_videos=torch.from_numpy(np.concatenate([pxs[0], pxs[0]])),
                               video_grid_thw=np.asarray([grid] * 2), second_per_grid_ts=[1.0] * 2).numpy()
    assert rel_err(wrong, want[2:]).max() > 1e-2
    out = m.generate(input_ids=ids, attention_mask=am, pixel_values_videos=pvt, video_grid_thw=gtt, second_per_grid_ts=[1.0] * 4,
                     generation_config=GenerationConfig(max_new_tokens=4, eos_token_id=-1, seed=3))
    assert out.shape == (4, ids.shape[1] + 4)
    # nucleus / min-p / repetition penalty ride on the top-k candidate path (HF's warper order); outside it they are refused
    out_p = m.generate(input_ids=ids, attention_mask=am, pixel_values_videos=pvt, video_grid_thw=gtt, second_per_grid_ts=[1.0] * 4,
                       generation_config=GenerationConfig(max_new_tokens=4, eos_token_id=-1, seed=3, top_p=0.9, min_p=0.02, repetition_penalty=1.2))
    assert out_p.shape == (4, ids.shape[1] + 4)
    with pytest.raises(ValueError, match="top_k"):
        m.generate(input_ids=ids, attention_mask=am, pixel_values_videos=pvt, video_grid_thw=gtt, second_per_grid_ts=[1.0] * 4,
                   generation_config=GenerationConfig(max_new_tokens=4, top_p=0.9, top_k=200))


def test_generate_with_hf_keyword_arguments_as_the_eval_harness_calls_it():
    """lmms-eval's model wrapper calls model.generate(**inputs, eos_token_id=, pad_token_id=, do_sample=False, temperature=None,
    top_p=None, num_beams=1, max_new_tokens=, use_cache=) (eval/lmms-eval/lmms_eval/models/qwen2_5_vl.py:326-336): greedy decode,
    deterministic, equal to the argmax of the oracle's teacher-forced logits; HF-only options are refused, not ignored."""
    m, w, ocfg = _model(2, 43)
    cfg = m.cfg
    g = torch.Generator().manual_seed(4)
    frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    tv = grid[0] * grid[1] * grid[2] // 4
    p = [21, cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 30, 31, 32]
    ids = torch.tensor([p])
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=torch.from_numpy(px).cuda(),
              video_grid_thw=torch.tensor([grid]), second_per_grid_ts=[1.0])
    C = 6
    out = m.generate(**kw, eos_token_id=-1, pad_token_id=cfg.pad_token_id, do_sample=False, temperature=None, top_p=None, num_beams=1,
                     max_new_tokens=C, use_cache=True)
    out2 = m.generate(**kw, eos_token_id=-1, pad_token_id=cfg.pad_token_id, do_sample=False, temperature=None, top_p=None, num_beams=1,
                      max_new_tokens=C, use_cache=True)
    assert out.shape == (1, ids.shape[1] + C) and torch.equal(out, out2)
    full = out.cpu()
    logits = om.forward_logits(w, ocfg, full, np.ones_like(full.numpy()), torch.from_numpy(px), np.asarray([grid]), [1.0])
    S = ids.shape[1]
    top2 = logits[0, S - 1:S + C - 1].topk(2, -1)
    for j in range(C):                                           # the chosen token is the oracle's argmax (unless the top two are a near tie)
        tok = int(full[0, S + j])
        assert tok == int(top2.indices[j, 0]) or float(top2.values[j, 0] - top2.values[j, 1]) < 2e-2, j
    with pytest.raises(ValueError, match="beam"):
        m.generate(**kw, num_beams=4, max_new_tokens=2)
    # an HF-shaped generation_config object, as the reference trainer passes it (transformers.GenerationConfig built at
    # grpo_trainer.py:559-570: HF's attribute names, a LIST of eos ids, no `seed` attribute): read, never modified
    from types import SimpleNamespace
    hf_gc = SimpleNamespace(max_new_tokens=C, do_sample=False, temperature=0.9, top_p=1.0, top_k=50, min_p=None, repetition_penalty=1.0,
                            pad_token_id=cfg.pad_token_id, eos_token_id=[-1], num_beams=1, max_length=None)
    out3 = m.generate(**kw, generation_config=hf_gc)
    assert torch.equal(out3, out) and hf_gc.eos_token_id == [-1]
    with pytest.raises(ValueError, match="beam"):
        m.generate(**kw, generation_config=SimpleNamespace(max_new_tokens=2, do_sample=False, num_beams=2))
    # sampling with HF's warpers: every drawn token lies in the set the oracle's warpers keep for ITS step (teacher-forced oracle on
    # the engine's own tokens); a repetition penalty > 1 is applied to prompt + generated ids before them
    outs = m.generate(**kw, eos_token_id=-1, pad_token_id=cfg.pad_token_id, do_sample=True, temperature=0.7, top_k=20, top_p=0.8, min_p=0.05,
                      repetition_penalty=1.3, max_new_tokens=C)
    fulls = outs.cpu()
    lg = om.forward_logits(w, ocfg, fulls, np.ones_like(fulls.numpy()), torch.from_numpy(px), np.asarray([grid]), [1.0])
    for j in range(C):
        raw = om.repetition_penalty(lg[0, S - 1 + j][None], fulls[:, :S + j], 1.3)
        kept = om.warp_logits(raw, 0.7, 20, 0.8, 0.05)[0]
        tok = int(fulls[0, S + j])
        near = (kept.exp().sum() > 0) and bool(torch.isfinite(kept[tok]))
        if not near:   # bf16-level logit noise can move a token across a warper's boundary: accept it within 2e-2 of the kept set's edge
            edge = kept[torch.isfinite(kept)].min()
            assert float(edge - raw[0, tok] / 0.7) < 5e-2, (j, tok)
    with pytest.raises(ValueError, match="do not match"):          # image features without image tokens (HF's masked_scatter check)
        m.generate(**kw, pixel_values=torch.zeros(4, 1176), max_new_tokens=2)


def test_generate_without_a_seed_draws_one_per_call_from_the_global_generator():
    """HF samples from torch's global RNG: two generate() calls with the same transformers.GenerationConfig (which has no
    `seed`) give different rollouts, torch.manual_seed() repeats them.  The engine's counter-based stream is keyed per call by a
    draw from the global CPU generator; an explicit seed pins it."""
    from types import SimpleNamespace
    from video_r2_amd.model import GenerationConfig
    m, w, ocfg = _model(2, 47)
    cfg = m.cfg
    g = torch.Generator().manual_seed(6)
    frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    tv = grid[0] * grid[1] * grid[2] // 4
    p = [21, cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 30, 31, 32]
    ids = torch.tensor([p, p, p, p])
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values_videos=torch.from_numpy(np.concatenate([px] * 4)).cuda(),
              video_grid_thw=torch.tensor([grid] * 4), second_per_grid_ts=[1.0] * 4)
    hf_gc = SimpleNamespace(max_new_tokens=12, do_sample=True, temperature=1.0, top_p=1.0, top_k=50, min_p=None, repetition_penalty=1.0,
                            pad_token_id=cfg.pad_token_id, eos_token_id=[-1], num_beams=1, max_length=None)
    torch.manual_seed(123)
    a = m.generate(**kw, generation_config=hf_gc)
    b = m.generate(**kw, generation_config=hf_gc)
    assert not torch.equal(a, b)                                   # 4 x 12 draws from 50 candidates each: a collision is impossible
    torch.manual_seed(123)
    a2 = m.generate(**kw, generation_config=hf_gc)
    assert torch.equal(a, a2)                                      # the global seed reproduces the run, as with HF
    c = m.generate(**kw, do_sample=True, temperature=1.0, top_k=50, max_new_tokens=12, eos_token_id=-1)
    d = m.generate(**kw, do_sample=True, temperature=1.0, top_k=50, max_new_tokens=12, eos_token_id=-1)
    assert not torch.equal(c, d)                                   # the keyword-argument form (the eval harness) too
    e = m.generate(**kw, generation_config=GenerationConfig(max_new_tokens=12, temperature=1.0, eos_token_id=-1, seed=9))
    f = m.generate(**kw, generation_config=GenerationConfig(max_new_tokens=12, temperature=1.0, eos_token_id=-1, seed=9))
    assert torch.equal(e, f)                                       # an explicit seed pins the stream


def test_save_pretrained_roundtrip(tmp_path):
    from video_r2_amd.model import VideoR2Model
    m, w, ocfg = _model(7, 23)
    m.save_pretrained(str(tmp_path))
    assert os.path.exists(tmp_path / "config.json") and os.path.exists(tmp_path / "model.safetensors")
    m2 = VideoR2Model.from_pretrained(str(tmp_path))
    sd1, sd2 = m.state_dict(), m2.state_dict()
    assert set(sd1) == set(w) and all(torch.equal(sd1[k].cpu(), sd2[k].cpu()) for k in sd1)
    assert all(torch.equal(sd1[k].float().cpu(), w[k]) for k in w)          # HF legacy layout, exact values


def test_trainer_step_metrics_and_loss():
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 29)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "12", "--beta", "0.04", "--loss_type", "grpo",
                              "--reward_func_names", "format_reward,accuracy_reward", "--reward_weights", "1.0,1.0"])
    g = torch.Generator().manual_seed(3)

    def processing_fn(ex):
        frames = torch.randint(0, 256, (2, 3, 56, 56), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).cuda(), video_grid_thw=[grid],
                    second_per_grid_ts=[1.0],
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if i % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], assistant="<answer>B</answer>", vid=i) for i in range(6)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    steps = list(tr.iter_steps())
    assert len(steps) == 6 and all(len(s) == 1 for s in steps)
    before = m.text["l0.o_w"].clone()
    out = tr.step(steps[0])
    assert np.isfinite(out["loss"]) and out["update"] is True and out["metrics"]["grad_norm"] > 0
    assert not torch.equal(before, m.text["l0.o_w"])          # the policy weights (bf16 compute copy) were updated in place
    ro = out["rollouts"][0]
    assert ro["completion_ids"].shape == (4, 12) and ro["logps"].shape == (4, 12) and ro["dlogp"].shape == (4, 12)
    # policy == reference copy at step 0 -> KL exactly 0, ratio exactly 1 -> loss = -mean_g(mean_t(A)) = -mean(A)
    adv = out["advantages"]["advantages"]
    # (EXACTLY: the reference's log-probs come from Engine.score_unified, the row space and launches of the training forward)
    assert out["metrics"]["kl"] == 0.0 and abs(out["loss"] - (-adv.mean().item())) < 1e-5
    assert torch.equal(ro["ref_logps"], ro["logps"])
    for k in ("completions/mean_length", "completions/clipped_ratio", "reward", "reward_std", "rewards/format_reward/mean",
              "rewards/accuracy_reward/mean", "clip_ratio/low_mean", "clip_ratio/high_max", "clip_ratio/region_mean", "num_tokens"):
        assert k in out["metrics"], k
    # phase times under the reference's metric names (profiling_decorator / profiling_context, grpo_trainer.py:711-722,850): HIP events
    # on the engine stream for the device phases, wall clock for the reward functions
    pt = {k.split(".")[-1]: v for k, v in out["metrics"].items() if k.startswith("profiling/Time taken: QwenGRPOTrainer.")}
    assert set(pt) == {"generate", "_get_per_token_logps", "backward", "optimizer_step", "format_reward", "accuracy_reward"}, pt
    assert all(v >= 0 for v in pt.values()) and pt["generate"] > 0 and pt["_get_per_token_logps"] > 0 and pt["backward"] > 0


def test_sync_ref_model_mixes_the_policy_into_the_reference():
    """--sync_ref_model (TRL's SyncRefModelCallback behind grpo_trainer.py:587-588): after every ref_model_sync_steps optimiser
    steps ref <- alpha * policy + (1 - alpha) * ref, on every LM tensor; off by default (the reference stays the SFT model)"""
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 37)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "8", "--beta", "0.04", "--learning_rate", "1e-2",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--sync_ref_model", "True",
                              "--ref_model_mixup_alpha", "0.6", "--ref_model_sync_steps", "2"])

    def processing_fn(ex):
        frames = torch.randint(0, 256, (2, 3, 56, 56), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).cuda(), video_grid_thw=[grid], second_per_grid_ts=[1.0],
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if i % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], assistant="<answer>B</answer>", vid=i) for i in range(4)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    steps = list(tr.iter_steps())
    ref0 = {k: v.clone() for k, v in tr.ref_params.t.items()}
    tr.step(steps[0])
    assert all(torch.equal(tr.ref_params.t[k], ref0[k]) for k in ref0)              # step 1: not a sync step
    tr.step(steps[1])
    m.engine.sync()
    moved = 0
    for k in ("l0.qkv_w", "l1.down_w", "lm_head", "embed", "l0.ln1", "norm", "l1.qkv_b"):
        want = 0.4 * ref0[k].float() + 0.6 * m.text.t[k].float()
        err = (tr.ref_params.t[k].float() - want).abs().max().item()
        assert err <= 2 ** -8 * want.abs().max().item(), (k, err)                    # one bf16 rounding of the fp32 mix
        moved += int(not torch.equal(tr.ref_params.t[k], ref0[k]))
    assert moved >= 4                                                                # the policy had moved (lr 1e-2), so did the reference


def _dp_worker(rank, world, port, q, backend="gloo", wire="fp32", vision=False):
    import os
    local = rank if backend == "nccl" else 0                     # RCCL wants one GPU per rank; the gloo rehearsal shares GPU 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
                      VR2_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from video_r2_amd import dist as vdist, params, rewards
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.trainer import QwenGRPOTrainer
    torch.cuda.set_device(local)
    vdist.init(device=f"cuda:{local}")
    ocfg = om.cfg_tiny(2)
    w = om.make_weights(ocfg, seed=29)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device=f"cuda:{local}")
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "8", "--beta", "0.04", "--loss_type", "bnpo",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--learning_rate", "1e-3",
                              "--grad_exchange_dtype", wire])
    args.train_vision_tower = args.train_merger = vision         # as train_grpo.py's check_trainable_set sets them

    def processing_fn(ex):
        frames = torch.randint(0, 256, (2, 3, 56, 56), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).to(m.device), video_grid_thw=[grid], second_per_grid_ts=[1.0],
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if (i + ex["vid"]) % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(8)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    steps = list(tr.iter_steps())
    out = tr.step(steps[0])
    torch.cuda.synchronize()
    assert tr.updater._works == {}                               # every bucket was exchanged and waited for
    q.put((rank, [ex["vid"] for ex in steps[0]], out["metrics"]["loss"], out["metrics"]["grad_norm"],
           m.text["l0.o_w"].float().sum().item(), m.text["l1.ln2"].sum().item() + m.vision["b1.qkv_w"].float().sum().item()
           + m.vision["fc2_b"].sum().item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16", "fp32+vision", "bf16+vision"])
def test_two_rank_dp_step_keeps_replicas_identical(wire):
    """world_size 2 on the one GPU (gloo collectives): ranks take different prompt groups, all-reduce the gradients and must end
    the step with bit-identical weights and identical reduced metrics -- with the fp32 arena on the wire and with the bf16 wire
    format (--grad_exchange_dtype bf16: half the bytes, every rank receives the same reduced values)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    vision = wire.endswith("+vision")                            # the tower and the merger train too: their bucket is exchanged as well
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, "gloo", wire.split("+")[0], vision)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, v0, l0, g0, w0, n0), (r1, v1, l1, g1, w1, n1) = res
    assert v0 != v1 and len(v0) == len(v1) == 1                  # different prompt groups per rank
    assert l0 == l1 and g0 == g1 and g0 > 0                      # reduced metrics agree; the clipped norm is of the MEAN gradient
    assert w0 == w1 and n0 == n1                                 # replicas stay bit-identical after the update


def _split_group_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      VR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from video_r2_amd import dist as vdist, params, rewards
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.trainer import QwenGRPOTrainer
    torch.cuda.set_device(0)
    vdist.init(device="cuda:0")
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=29), device="cuda:0")
    cfg = m.cfg
    # the shipped recipe's shape in small: FEWER completions per device than a group holds (2 of G = 4 on each of 2 ranks)
    args = params.parse_args(["--num_generations", "4", "--per_device_train_batch_size", "2", "--max_completion_length", "8", "--beta", "0.04",
                              "--loss_type", "bnpo", "--reward_func_names", "length_parity_reward", "--reward_weights", "1.0",
                              "--learning_rate", "1e-3"])

    def processing_fn(ex):
        ids = [40 + ex["vid"], 50, 51, 52]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None,
                    decode=lambda toks: [" ".join(str(int(t)) for t in row) for row in toks.cpu().numpy()])

    def length_parity_reward(completions, **kwargs):               # depends on the sampled tokens: differs between the ranks' slices
        return [float(sum(int(t) for t in c[0]["content"].split()) % 5) for c in completions]

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(6)]
    tr = QwenGRPOTrainer(m, [length_parity_reward], args, data, processing_fn, reward_weights=[1.0], reward_func_names=["length_parity_reward"])
    assert tr.group_ranks == 2 and tr.local_generations == 2 and tr.groups_per_step == 1
    steps = list(tr.iter_steps())
    out = tr.step(steps[0])
    torch.cuda.synchronize()
    ro = out["rollouts"][0]
    q.put((rank, [ex["vid"] for ex in steps[0]], len(steps), ro["completion_ids"].cpu().tolist(), out["advantages"]["rewards"].tolist(),
           out["advantages"]["advantages"].tolist(), out["metrics"]["loss"], out["metrics"]["grad_norm"], out["metrics"]["reward"],
           m.text["l0.o_w"].float().sum().item(), m.text["l1.ln2"].sum().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_one_prompt_group_spread_over_two_ranks_like_the_reference_recipe():
    """per_device_train_batch_size < num_generations (train_grpo.sh: 1 per device, G = 8, 8 devices): the reference deals the G
    repeats of a prompt to G / per_device ranks; rewards are gathered and the advantages taken over the whole group
    (grpo_trainer.py:894-981).  Here: world 2, G = 4, 2 completions per rank -- both ranks work on the SAME prompt, sample different
    completions, the advantages of the union are the oracle's group advantages, the step takes ONE prompt (not `world`), and the
    replicas end the step bit-identical."""
    import socket
    import torch.multiprocessing as mp
    from oracle import grpo as og
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_split_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, v0, n0, c0, rw0, a0, l0, g0, m0, w0, x0), (r1, v1, n1, c1, rw1, a1, l1, g1, m1, w1, x1) = res
    assert v0 == v1 and len(v0) == 1 and n0 == n1 == 6             # the same prompt on both ranks; 6 prompts -> 6 steps (not 3)
    assert c0 != c1 and len(c0) == len(c1) == 2                     # each rank sampled its own 2 of the 4 completions
    want = og.advantages(np.asarray(rw0 + rw1, dtype=np.float32)[:, None], [1.0], 4)["advantages"]
    assert np.allclose(np.asarray(a0 + a1), want, atol=1e-5)        # advantages over the WHOLE group, each rank holding its rows
    assert l0 == l1 and g0 == g1 and m0 == m1                       # reduced metrics agree (reward = the group's mean on both)
    assert w0 == w1 and x0 == x1                                    # replicas stay bit-identical after the update


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path needs two GPUs (the gloo rehearsal above runs the same code on one)")
def test_two_rank_dp_step_on_rccl():
    """the same step on the "nccl" backend (RCCL over xGMI), one GPU per rank: per-layer gradient buckets all-reduced on RCCL's
    stream under the backward, packed metrics all-reduce on the device, replicas bit-identical afterwards"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, "nccl")) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, v0, l0, g0, w0, n0), (r1, v1, l1, g1, w1, n1) = res
    assert v0 != v1 and l0 == l1 and g0 == g1 and g0 > 0 and w0 == w1 and n0 == n1


def test_checkpoint_resume_and_num_iterations(tmp_path):
    """two optimiser steps with num_iterations=2 (second step re-uses the buffered rollout with old-policy log-probs), then
    save -> fresh trainer -> load: masters, moments, compute copies and counters are restored exactly."""
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 29)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "8", "--num_iterations", "2", "--learning_rate", "1e-3",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--loss_type", "bnpo"])

    def processing_fn(ex):
        ids = [40 + ex["vid"], 50, 51, 52]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None,
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if i % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(4)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    steps = list(tr.iter_steps())
    assert len(steps) == 8 and steps[0][0]["vid"] == steps[1][0]["vid"] != steps[2][0]["vid"]     # each prompt twice (num_iterations)
    o1 = tr.step(steps[0])
    o2 = tr.step(steps[1])
    assert o2["rollouts"] is o1["rollouts"] and "old_logps" in o2["rollouts"][0]                  # buffered rollout re-used
    assert o2["metrics"]["clip_ratio/region_mean"] >= 0 and tr.global_step == 2
    d = tr.save_checkpoint(str(tmp_path))
    assert os.path.exists(os.path.join(d, "model.safetensors")) and os.path.exists(os.path.join(d, "trainer_state.pt"))
    from video_r2_amd.model import VideoR2Model
    m2 = VideoR2Model.from_pretrained(d)
    tr2 = QwenGRPOTrainer(m2, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names).load_checkpoint(d)
    assert tr2.global_step == 2 and tr2.updater.state.step_count == 2
    for k in tr.updater.state.master:
        assert torch.equal(tr.updater.state.master[k], tr2.updater.state.master[k]), k
        assert torch.equal(tr.updater.state.m[k], tr2.updater.state.m[k]) and torch.equal(tr.updater.state.v[k], tr2.updater.state.v[k])
    for k in ("l0.qkv_w", "l1.down_w_t", "lm_head", "embed"):
        assert torch.equal(m.text[k], m2.text[k]), k
    assert len(list(tr2.iter_steps())) == 6                                                      # the two finished steps are skipped


def test_config5_old_policy_logps_come_from_the_bf16_arithmetic_of_the_training_forward():
    """--decode_weights fp8 (config 5) with --num_iterations 2: the rollout decodes and the REFERENCE scores on fp8 operands, but the
    old-policy log-probs that the ratio is taken against are computed by the bf16 GEMMs of the training forward (Engine.score_unified
    allow_fp8=False) -- at iteration 0 they equal the policy's log-probs bit for bit, the ratio is exactly 1 and no clip branch fires,
    as in the reference (grpo_trainer.py:1072-1077: old = logps.detach() on-policy)."""
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 53)
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "8", "--num_iterations", "2", "--learning_rate", "1e-3",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--loss_type", "bnpo", "--beta", "0.04",
                              "--decode_weights", "fp8"])

    def processing_fn(ex):
        ids = [40 + ex["vid"], 50, 51, 52]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=None, video_grid_thw=None, second_per_grid_ts=None,
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if i % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(4)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    assert m.engine.gemm_fp8 and m.engine.decode_fp8
    steps = list(tr.iter_steps())
    o1 = tr.step(steps[0])
    ro = o1["rollouts"][0]
    assert torch.equal(ro["old_logps"], ro["logps"])                              # bf16 pass == bf16 training forward, bit for bit
    assert o1["metrics"]["clip_ratio/region_mean"] == 0.0 and o1["metrics"]["clip_ratio/low_mean"] == 0.0 and o1["metrics"]["clip_ratio/high_mean"] == 0.0
    assert not torch.equal(ro["ref_logps"], ro["logps"])                          # the reference DID run on fp8 operands
    assert m.engine.gemm_fp8                                                      # (the mode is restored after the bf16 pass)


def test_bench_two_rank_rehearsal_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (rank-dependent prompts, barriers, max-over-ranks timing, the gradient all-reduce in EVERY step
    including the instrumented one) with 2 ranks on this one GPU over gloo and the tiny config: every rank must issue the same
    collectives or the job dies -- the real N = 2/4/8 runs use the same code with RCCL."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VR2_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--model", "tiny", "--frames", "4", "--G", "2",
           "--max-new", "12", "--steps", "2", "--warmup", "1", "--prompts-per-gpu", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and d["roofline_decode"]["bound"] == "hbm"     # dominant kernel of the step / of the decode
    assert "cpu_baseline" not in d                                          # the CPU baseline is an N = 1 leg only
    assert d["eos_mix"]["samples_per_s"] > 0                                # the emulated-EOS step pair (packed rows) ran on every rank


def test_trainer_batches_the_rank_groups_in_one_decode_and_keeps_them_apart():
    """several prompt groups per rank and step: encoded one by one, decoded TOGETHER (shared-stride prefix slots), rewards on
    a background thread -- every group's reference / policy log-probs must be ITS OWN (the scoring outputs are reused
    workspaces) and equal to a fresh scoring of that group's tokens alone; sync and async reward paths agree"""
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 31)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "10", "--beta", "0.04", "--loss_type", "bnpo",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--per_device_train_batch_size", "12"])
    seen = []

    def processing_fn(ex):
        hw = (56, 56) if ex["vid"] % 2 else (56, 84)                   # prompt lengths differ between the groups
        frames = torch.randint(0, 256, (2, 3, *hw), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]

        def decode(toks):
            seen.append(int(ex["vid"]))
            return ["<think>t</think> <answer>B</answer>" if (i + ex["vid"]) % 2 else "nope" for i in range(toks.shape[0])]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).cuda(), video_grid_thw=[grid],
                    second_per_grid_ts=[1.0], decode=decode)

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(6)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    tr.update = False                                                  # rollout + scoring only: weights stay put for the re-scoring below
    examples = [data[i] for i in (0, 1, 2)]
    rollouts, rew = tr._rollout_all(examples, seed=5)
    assert len(rollouts) == 3 and rew.shape == (12, 1) and sorted(seen) == [0, 1, 2]
    eng = m.engine
    for ex, ro in zip(examples, rollouts):
        pi = processing_fn(ex)
        emb, plan = eng.encode_video(pi["pixel_values_videos"], pi["video_grid_thw"])
        st = eng.prefill(m.text, ro["prompt_ids"], emb, plan, [1.0])
        lp = eng.score(m.text, st, ro["completion_ids"], args.temperature)
        eng.sync()
        assert torch.equal(lp, ro["logps"]) and torch.equal(lp, ro["ref_logps"])   # reference == policy copy at step 0
    assert not torch.equal(rollouts[0]["logps"], rollouts[1]["logps"])
    args.async_rewards = False
    seen.clear()
    rollouts2, rew2 = tr._rollout_all(examples, seed=5)
    assert torch.equal(rew, rew2) and all(torch.equal(a["completion_ids"], b["completion_ids"]) for a, b in zip(rollouts, rollouts2))


def test_trainer_step_with_fp8_decode_weights():
    """--decode_weights fp8: the rollout decodes on e4m3 weight copies, the update runs on the fp32 masters / bf16 copies, and
    the fp8 copies are re-quantised IN PLACE after the optimiser step (captured decode graphs keep their addresses)"""
    from video_r2_amd import params, rewards
    from video_r2_amd.trainer import QwenGRPOTrainer
    m, w, ocfg = _model(2, 37)
    cfg = m.cfg
    args = params.parse_args(["--num_generations", "4", "--max_completion_length", "10", "--beta", "0.04", "--loss_type", "bnpo",
                              "--reward_func_names", "format_reward", "--reward_weights", "1.0", "--learning_rate", "1e-2",
                              "--decode_weights", "fp8", "--per_device_train_batch_size", "8"])

    def processing_fn(ex):
        frames = torch.randint(0, 256, (2, 3, 56, 56), generator=torch.Generator().manual_seed(ex["vid"]), dtype=torch.uint8).numpy()
        px, grid = pixels_from_frames(frames, cfg.vision)
        tv = grid[0] * grid[1] * grid[2] // 4
        ids = [40 + ex["vid"], cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id, 50, 51]
        return dict(input_ids=np.asarray(ids), pixel_values_videos=torch.from_numpy(px).cuda(), video_grid_thw=[grid], second_per_grid_ts=[1.0],
                    decode=lambda toks: ["<think>t</think> <answer>B</answer>" if (i + ex["vid"]) % 2 else "nope" for i in range(toks.shape[0])])

    data = [dict(prompt=[{"role": "user", "content": "q"}], vid=i) for i in range(4)]
    funcs, wts, names = rewards.load_reward_funcs("video_r2_amd.rewards", args)
    tr = QwenGRPOTrainer(m, funcs, args, data, processing_fn, reward_weights=wts, reward_func_names=names)
    assert m.engine.decode_fp8 and "l0.gu_w" in m.text.f8
    q_before, s_before = (t.clone() for t in m.text.f8["l0.down_w"])
    ptr = m.text.f8["l0.down_w"][0].data_ptr()
    out = tr.step([data[0], data[1]])
    assert np.isfinite(out["loss"]) and out["metrics"]["grad_norm"] > 0
    q_after, s_after = m.text.f8["l0.down_w"]
    assert q_after.data_ptr() == ptr and (not torch.equal(q_after, q_before) or not torch.equal(s_after, s_before))
    out2 = tr.step([data[2], data[3]])                              # second rollout replays the graphs on the refreshed copies
    assert np.isfinite(out2["loss"])


def _rccl_single_rank_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from video_r2_amd import lib
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda:0"))
    lib.load()                                                   # libvr2.so and RCCL share one HIP runtime in this process
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        y = torch.zeros(4, 512, dtype=torch.bfloat16, device="cuda:0")
        lib.call("vr2_cast_rows_f32_bf16", x.view(2048, 512), 512, None, y, 512, 4, 512, 512, st.cuda_stream)
        wk = dist.all_reduce(x, async_op=True)                   # ordered after the kernel above, runs on RCCL's stream
        wk.wait()
        got = [torch.zeros(1, dtype=torch.float64, device="cuda:0")]
        dist.all_gather(got, torch.tensor([3.5], dtype=torch.float64, device="cuda:0"))
    dist.barrier()
    torch.cuda.synchronize()
    q.put((float(x[12345].item()), float(got[0].item()), float(y[0, 7].item())))
    dist.destroy_process_group()


def test_rccl_loads_and_runs_collectives_next_to_the_engine_library():
    """the "nccl" backend (= RCCL) in a process that also runs libvr2.so kernels: process-group init with device_id, an async
    all-reduce ordered behind an engine-stream kernel, all_gather of the timing scalar and a barrier -- the calls bench.py and
    train.py make for N > 1, here on the one GPU of the test box (world size 1: the multi-rank path needs an 8-GPU node)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(port, q))
    p.start()
    a, b, c = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0 and a == 12345.0 and b == 3.5 and c == 7.0


def test_facade_takes_image_inputs_like_hf(golden_dir):
    """generate / per_token_logps with the `images=` half of the processor output (pixel_values, image_grid_thw) beside the video
    half, as the reference forwards **prompt_inputs (grpo_trainer.py:759-786, :723-741): log-probs against the HF golden vectors,
    copies of one sample recognised as a group, greedy generate == a direct engine decode of the same prompt."""
    import os
    from oracle import model as om
    from tests.helpers import to_engine_cfg, pixels_from_frames, rel_err
    from video_r2_amd import weights as W
    from video_r2_amd.model import VideoR2Model
    for name in ("fwd_tiny_g2_img", "fwd_tiny_g2_img2"):
        f = np.load(os.path.join(golden_dir, name + ".npz"))
        ocfg = om.cfg_tiny(int(f["group"]))
        cfg = to_engine_cfg(ocfg)
        m = VideoR2Model.from_state_dict(cfg, om.make_weights(ocfg, seed=int(f["seed"])))
        C = int(f["C"])
        ids, am = torch.from_numpy(f["input_ids"]), torch.from_numpy(f["attention_mask"])
        mm = dict(pixel_values=torch.from_numpy(f["pixel_values"]), image_grid_thw=torch.from_numpy(f["image_grid_thw"]))
        if "frames_u8" in f:
            px, grid = pixels_from_frames(f["frames_u8"], cfg.vision)
            mm.update(pixel_values_videos=torch.from_numpy(px), video_grid_thw=torch.tensor([grid]), second_per_grid_ts=[float(f["sec"])])
        # two copies of the sample in one batch (the reference batches G copies): every tensor repeated per row
        rep = {k: (torch.cat([v, v], 0) if torch.is_tensor(v) else list(v) * 2) for k, v in mm.items()}
        lp = m.per_token_logps(torch.cat([ids, ids], 0), torch.cat([am, am], 0), C, temperature=float(f["temperature"]), **rep)
        m.engine.sync()
        for b in range(2):
            r = rel_err(lp[b].cpu().numpy(), f["hf_logps"][0])
            assert r.max() < 1e-3, (name, b, r.max())
        P = ids.shape[1] - C
        out = m.generate(ids[:, :P], am[:, :P], max_new_tokens=6, do_sample=False, **mm)
        assert out.shape == (1, P + 6) and bool((out[:, :P].cpu() == ids[:, :P]).all())
        # the same prompt through the engine directly
        media = m._media(ids[:, :P].numpy(), am[:, :P].numpy(), mm["pixel_values"], mm["image_grid_thw"], mm.get("pixel_values_videos"),
                         mm.get("video_grid_thw"), mm.get("second_per_grid_ts"))
        emb, plan, sec = m._encode_group(media, 0)
        st = m.engine.prefill(m.text, ids[0, :P][am[0, :P].bool()].numpy(), emb, plan, sec)
        toks, _ = m.engine.decode(m.text, st, 1, 6, temperature=1.0, top_k=1, seed=0, eos_token_id=cfg.eos_token_id)
        m.engine.sync()
        assert bool((out[0, P:].cpu() == toks[0].cpu().long()).all())
    # image tokens without image features: HF's error, not a silent text prompt
    with pytest.raises(ValueError, match="do not match|no image pixel values"):
        m.per_token_logps(ids, am, C, image_grid_thw=None, pixel_values=None)


def test_forward_logits_serves_the_reference_logprob_body_unchanged(golden_dir):
    """`model(input_ids=, attention_mask=, **multimodal_inputs).logits` -- what the reference's _get_per_token_logps reads
    (grpo_trainer.py:725-739) -- followed by ITS slicing / temperature / gather - logsumexp: the HF golden log-probs come out,
    for a left-padded video prompt and for a prompt with still images."""
    import os
    from oracle import model as om
    from tests.helpers import to_engine_cfg, pixels_from_frames, rel_err
    from video_r2_amd.model import VideoR2Model
    for name in ("fwd_tiny_g2_pad", "fwd_tiny_g2_img"):
        f = np.load(os.path.join(golden_dir, name + ".npz"))
        ocfg = om.cfg_tiny(int(f["group"]))
        cfg = to_engine_cfg(ocfg)
        m = VideoR2Model.from_state_dict(cfg, om.make_weights(ocfg, seed=int(f["seed"])))
        C = int(f["C"])
        input_ids, attention_mask = torch.from_numpy(f["input_ids"]), torch.from_numpy(f["attention_mask"])
        px, grid = pixels_from_frames(f["frames_u8"], cfg.vision)
        mm = dict(pixel_values_videos=torch.from_numpy(px), video_grid_thw=torch.tensor([grid]), second_per_grid_ts=[float(f["sec"])])
        if "pixel_values" in f:
            mm.update(pixel_values=torch.from_numpy(f["pixel_values"]), image_grid_thw=torch.from_numpy(f["image_grid_thw"]))
        # ---- the reference's lines, verbatim in structure
        logits = m(input_ids=input_ids.to(m.device), attention_mask=attention_mask.to(m.device), **mm).logits
        logits = logits[:, :-1, :]
        ids = input_ids[:, 1:].to(m.device)
        logits = logits[:, -C:]
        ids = ids[:, -C:]
        logits = logits / float(f["temperature"])
        logps = torch.gather(logits, -1, ids.unsqueeze(-1)).squeeze(-1) - torch.logsumexp(logits, dim=-1)
        r = rel_err(logps[0].cpu().numpy(), f["hf_logps"][0])
        assert r.max() < 1e-3, (name, r.max())
        last = m(input_ids=input_ids, attention_mask=attention_mask, **mm).logits[0, -1]
        assert np.abs(last.cpu().numpy() - f["hf_last_logits"]).max() < 2e-2
