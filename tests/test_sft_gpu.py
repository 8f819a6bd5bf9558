"""GPU: the stage-1 SFT step on the engine (video_r2_amd/sft.py) against torch autograd through the fp32 CPU oracle: the supervised
loss of the reference's mixed-modality forward (oracle/sft.py: shifted token-mean cross entropy over a micro-batch) and the gradient
of every trainable tensor, rows with a video (two answer turns) and without; the schedule and the per-group learning rates."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import sft as osft
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg
from tests.test_train_gpu import _engine_grads_as_hf
from video_r2_amd import params

pytestmark = pytest.mark.gpu


def _rows(cfg, seed, n_text=1):
    """processed rows as SFTFrontEnd hands them over: one video row with two supervised answers, n_text text-only rows"""
    g = torch.Generator().manual_seed(seed)
    frames = torch.randint(0, 256, (4, 3, 56, 84), generator=g, dtype=torch.uint8)
    px, grid = pixels_from_frames(frames.numpy(), cfg.vision)
    tv = grid[0] * grid[1] * grid[2] // 4
    r = lambda n: torch.randint(10, 2000, (n,), generator=g).tolist()      # noqa: E731
    pieces = [(r(5), False), ([cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id] + r(6), False),
              (r(7) + [cfg.eos_token_id], True), (r(4), False), (r(5) + [cfg.eos_token_id], True)]
    ids, labels = osft.labels_of(pieces)
    rows = [dict(input_ids=ids, labels=labels, videos=[(frames, None)], second_per_grid_ts=[1.0], _px=px, _grid=grid)]
    for _ in range(n_text):
        ids, labels = osft.labels_of([(r(9), False), (r(6) + [cfg.eos_token_id], True)])
        rows.append(dict(input_ids=ids, labels=labels, videos=[], second_per_grid_ts=None, _px=None, _grid=None))
    return rows


def _oracle_step(ocfg, w, rows, trainable, lora=None, extra_params=()):
    """one micro-batch through the oracle: token-mean CE over the rows' supervised positions; gradients by autograd"""
    wt = {k: v.clone().requires_grad_(bool(trainable(k))) for k, v in w.items()}
    n = sum(int((r["labels"][1:] != -100).sum()) for r in rows)
    total = 0.0
    for r in rows:
        ids = torch.from_numpy(r["input_ids"])[None]
        px = None if r["_px"] is None else torch.from_numpy(r["_px"])
        grid = None if r["_grid"] is None else np.asarray([r["_grid"]])
        logits = om.forward_logits(wt, ocfg, ids, None, px, grid, r["second_per_grid_ts"], lora=None if lora is None else lora(r))
        lb = torch.from_numpy(r["labels"])[None]
        nll = torch.nn.functional.cross_entropy(logits[0, :-1].float(), lb[0, 1:], ignore_index=-100, reduction="sum")
        total = total + nll / n
    if any(v.requires_grad for v in wt.values()) or lora is not None:
        total.backward()
    return float(total.detach()), {k: v.grad for k, v in wt.items() if v.requires_grad and v.grad is not None}


def _trainer(ocfg, seed, argv, lora=None):
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    w = om.make_weights(ocfg, seed=seed)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    args = params.parse_sft_args(argv)
    rows = _rows(m.cfg, seed)
    tr = QwenSFTTrainer(m, args, list(range(8)), lambda ex: ex, lora=lora(m) if lora else None)
    return m, w, args, rows, tr


@pytest.mark.parametrize("group", [2, 7])
def test_sft_micro_batch_loss_and_gradients_vs_autograd(group):
    ocfg = om.cfg_tiny(group)
    m, w, args, rows, tr = _trainer(ocfg, 50 + group, ["--freeze_vision_tower", "True", "--freeze_merger", "True", "--per_device_train_batch_size", "2"])
    n_mb = sum(int((r["labels"][1:] != -100).sum()) for r in rows)
    parts = [tr._row(r, 1.0 / n_mb, last=(i == len(rows) - 1)) for i, r in enumerate(rows)]
    m.engine.sync()
    loss = sum(float(nll.item()) for nll, _ in parts) / n_mb
    assert sum(n for _, n in parts) == n_mb
    o_loss, o_grads = _oracle_step(ocfg, w, rows, lambda k: k.startswith("model.") or k == "lm_head.weight")
    assert abs(loss - o_loss) < 2e-3 * abs(o_loss), (loss, o_loss)
    got = _engine_grads_as_hf(tr.updater, m.cfg)
    worst = {k: (got[k] - g).norm().item() / max(g.norm().item(), 1e-12) for k, g in o_grads.items()}
    print("sft grads: loss %.5f vs %.5f; max rel grad err %.3e (median %.3e)" % (loss, o_loss, max(worst.values()), float(np.median(list(worst.values())))))
    assert set(got) == set(o_grads) and max(worst.values()) < 3e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:4]


def test_sft_step_schedule_accumulation_and_parameter_groups():
    """two optimiser steps of 2 accumulated micro-batches: the logged loss is the mean of the micro-batch losses, the learning rate
    follows HF's cosine-with-warm-up (0 at the first step with one warm-up step), and --vision_lr / --merger_lr move the vision tower
    and the merger at their own rates (first AdamW step: |delta| = lr for every element with a non-zero gradient)."""
    from video_r2_amd.train import cosine_lr
    ocfg = om.cfg_tiny(2)
    m, w, args, rows, tr = _trainer(ocfg, 61, ["--freeze_vision_tower", "False", "--freeze_merger", "False", "--learning_rate", "1e-3", "--vision_lr", "2e-4",
                                               "--merger_lr", "5e-4", "--gradient_accumulation_steps", "2", "--warmup_steps", "1", "--max_steps", "3",
                                               "--weight_decay", "0.0", "--max_grad_norm", "0"])
    assert tr.total_steps == 3 and tr.warmup_steps == 1 and tr.updater.vis is not None
    before = {k: v.clone() for k, v in list(tr.updater.state.master.items()) + list(tr.updater.vis.state.master.items())}
    o0 = tr.step([[0], [1]], [[rows[0]], [rows[1]]])
    assert o0["metrics"]["learning_rate"] == 0.0                                                   # HF LambdaLR: lambda(0) = 0 with warm-up
    for k, v in before.items():
        cur = tr.updater.state.master.get(k, tr.updater.vis.state.master.get(k))
        assert torch.equal(cur, v), k                                                              # lr 0: nothing moved (weight decay off)
    # the logged loss: mean over the micro-batches of each one's token mean
    l_video, _ = _oracle_step(ocfg, w, [rows[0]], lambda k: False)
    l_text, _ = _oracle_step(ocfg, w, [rows[1]], lambda k: False)
    assert abs(o0["loss"] - 0.5 * (l_video + l_text)) < 2e-3 * abs(o0["loss"])
    o1 = tr.step([[0], [1]], [[rows[0]], [rows[1]]])
    mult = cosine_lr(1.0, 1, 3, 1)
    assert mult == 1.0 and abs(o1["metrics"]["learning_rate"] - 1e-3) < 1e-9 and o1["metrics"]["grad_norm"] > 0
    st, vs = tr.updater.state, tr.updater.vis.state

    def moved(cur, old):
        d = (cur - old).abs()
        return float(d[d > 0].median())
    # Adam's second step after a zero-lr first step: m / sqrt(v) of two equal gradients is 1 up to eps -> |delta| = lr of the group
    assert abs(moved(st.master["l0.o_w"], before["l0.o_w"]) - 1e-3) < 2e-5
    assert abs(moved(vs.master["b0.qkv_w"], before["b0.qkv_w"]) - 2e-4) < 4e-6
    assert abs(moved(vs.master["fc2_w"], before["fc2_w"]) - 5e-4) < 1e-5
    assert tr.global_step == 2 and tr.num_input_tokens_seen == 2 * (len(rows[0]["input_ids"]) + len(rows[1]["input_ids"]))


# ------------------------------------------------------------------------------------------------ LoRA (the shipped SFT recipe)
def test_dropout_kernels_share_one_mask_and_keep_the_expected_share():
    from video_r2_amd import lib
    dev = "cuda:0"
    T, K, p = 96, 1024, 0.25
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    out, again, other = (torch.empty_like(x) for _ in range(3))
    lib.call("vr2_dropout_bf16", x, K, out, K, T, K, p, 7, 11, s)
    lib.call("vr2_dropout_bf16", x, K, again, K, T, K, p, 7, 11, s)
    lib.call("vr2_dropout_bf16", x, K, other, K, T, K, p, 7, 12, s)
    torch.cuda.synchronize()
    keep = out != 0
    assert torch.equal(out, again) and not torch.equal(out != 0, other != 0)                 # a pure function of (seed, salt, t, k)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    assert torch.equal(out[keep], (x.float() / (1 - p)).to(torch.bfloat16)[keep])               # torch.nn.Dropout's scaling
    rows, cols = keep.float().mean(1), keep.float().mean(0)
    assert rows.min() > 0.65 and rows.max() < 0.85 and cols.min() > 0.5 and cols.max() < 0.95  # no dead rows / columns
    g = torch.randn(T, K, device=dev)
    d32, d16 = torch.ones(T, K, device=dev), torch.ones(T, K, device=dev).to(torch.bfloat16)
    lib.call("vr2_dropout_add", g, K, d32, K, 0, T, K, p, 7, 11, s)
    lib.call("vr2_dropout_add", g, K, d16, K, 1, T, K, p, 7, 11, s)
    torch.cuda.synchronize()
    want = 1.0 + torch.where(keep, g / (1 - p), torch.zeros_like(g))
    assert torch.allclose(d32, want, atol=1e-6) and torch.equal(d16, want.to(torch.bfloat16))    # the backward's mask IS the forward's
    full = torch.empty_like(x)
    lib.call("vr2_dropout_bf16", x, K, full, K, T, K, 0.0, 7, 11, s)
    torch.cuda.synchronize()
    assert torch.equal(full, x)                                                                   # p = 0 keeps everything, unscaled


def _lora_pair(ocfg, m, w, r, alpha, p, b_std):
    """the same adapters on both sides: oracle tensors (bf16-representable) loaded into the engine in peft's layout"""
    from video_r2_amd.lora import LoraTrainer
    lo_o = osft.LoraOracle(w, ocfg.text.num_hidden_layers, r, alpha, seed=3, b_std=b_std)
    for d in (lo_o.A, lo_o.B):
        for k in d:
            d[k] = d[k].detach().to(torch.bfloat16).float().requires_grad_(True)
    lo_e = LoraTrainer(m.engine, m.text, rank=r, alpha=alpha, dropout=p, seed=5)
    sd = {}
    for k in lo_o.A:
        sd[f"base_model.model.{k}.lora_A.weight"] = lo_o.A[k].detach()
        sd[f"base_model.model.{k}.lora_B.weight"] = lo_o.B[k].detach()
    lo_e.load_adapter_state_dict(sd)
    return lo_o, lo_e


def _dropout_masks(lo_e, lo_o, rows, p, n_layers, first_fid=1):
    """per row: {HF module: 0/1 mask [1, T, in]} = the draws the engine's forward number first_fid + i will make (one per fused site,
    shared by the site's modules), taken from the dropout kernel itself on a block of ones"""
    from video_r2_amd import lib
    from video_r2_amd.lora import SITES
    site_mods = {"qkv": ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), "o": ("self_attn.o_proj",),
                 "gu": ("mlp.gate_proj", "mlp.up_proj"), "down": ("mlp.down_proj",)}
    s = torch.cuda.current_stream().cuda_stream
    out_all = []
    for fid, r in enumerate(rows, start=first_fid):
        T = len(r["input_ids"])
        mk = {}
        for l in range(n_layers):
            for site in SITES:
                K = lo_e.W.shape[f"l{l}.{site}.A"][1]
                ones, out = torch.ones(T, K, device="cuda:0", dtype=torch.bfloat16), torch.empty(T, K, device="cuda:0", dtype=torch.bfloat16)
                sd_, salt = lo_e.draw_keys(fid, l, site)
                lib.call("vr2_dropout_bf16", ones, K, out, K, T, K, p, sd_, salt, s)
                for mod in site_mods[site]:
                    fin = lo_o.A[f"model.layers.{l}.{mod}"].shape[1]
                    mk[f"model.layers.{l}.{mod}"] = (out != 0).float().cpu()[None, :, :fin]
        out_all.append(mk)
    torch.cuda.synchronize()
    return out_all


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_lora_sft_loss_and_adapter_gradients_vs_autograd(p):
    """base + B A (dropout(x)) * alpha / r on every LM linear layer, B != 0, against autograd through the oracle with the engine's own
    dropout draws; the frozen LM holds no gradients; peft-layout round trip; one optimiser step changes only the adapters"""
    from video_r2_amd.lora import SITES
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    ocfg = om.cfg_tiny(2)
    w = om.make_weights(ocfg, seed=77)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    args = params.parse_sft_args(["--freeze_llm", "True", "--freeze_vision_tower", "True", "--freeze_merger", "True", "--lora_enable", "True",
                                  "--lora_rank", "8", "--lora_alpha", "16", "--lora_dropout", str(p), "--lora_namespan_exclude", "['lm_head', 'embed_tokens']",
                                  "--learning_rate", "1e-3", "--weight_decay", "0.0"])
    lo_o, lo_e = _lora_pair(ocfg, m, w, 8, 16, p, 0.02)
    assert lo_e.n_params() == sum(a.numel() for a in lo_o.A.values()) + sum(b.numel() for b in lo_o.B.values())
    tr = QwenSFTTrainer(m, args, list(range(4)), lambda ex: ex, lora=lo_e)
    rows = _rows(m.cfg, 77)
    n_mb = sum(int((r["labels"][1:] != -100).sum()) for r in rows)
    masks_of = _dropout_masks(lo_e, lo_o, rows, p, ocfg.text.num_hidden_layers)
    parts = [tr._row(r, 1.0 / n_mb, last=(i == len(rows) - 1)) for i, r in enumerate(rows)]
    m.engine.sync()
    loss = sum(float(nll.item()) for nll, _ in parts) / n_mb
    assert tr.updater.state.master == {} and tr.updater.state.arena.numel() == 0         # the LM is frozen: adapters only

    def hooks(r):
        lo_o.set_dropout(masks_of[[id(x) for x in rows].index(id(r))] if p > 0 else {}, p)
        return lo_o.hooks()
    o_loss, _ = _oracle_step(ocfg, w, rows, lambda k: False, lora=hooks)
    assert abs(loss - o_loss) < 2e-3 * abs(o_loss), (loss, o_loss)
    lo_e.finish_grads()
    torch.cuda.synchronize()
    worst = {}
    for l in range(ocfg.text.num_hidden_layers):
        for site in SITES:
            gA, gB = lo_e.grad[(l, site, "A")], lo_e.grad[(l, site, "B")]
            assert float((gA * (1 - lo_e.mask[(l, site, "A")])).abs().max()) == 0.0 and float((gB * (1 - lo_e.mask[(l, site, "B")])).abs().max()) == 0.0
            for j, (mod, fin, rws) in enumerate(lo_e.layout[site]):
                name = f"model.layers.{l}.{mod}"
                ga = gA[j * lo_e.rp:j * lo_e.rp + 8, :fin].cpu()
                gb = gB[rws][:, j * lo_e.rp:j * lo_e.rp + 8].cpu()
                for got, ref, kind in ((ga, lo_o.A[name].grad, "A"), (gb, lo_o.B[name].grad, "B")):
                    worst[name + "." + kind] = (got - ref).norm().item() / max(ref.norm().item(), 1e-12)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("lora p=%.2f: loss %.5f vs %.5f; max rel grad err %.3e (median %.3e); worst %s"
          % (p, loss, o_loss, max(worst.values()), float(np.median(list(worst.values()))), ", ".join("%s %.2e" % kv for kv in top)))
    assert max(worst.values()) < 3e-2, top
    # peft layout round trip and the merge: W + s B A reproduces the adapted forward (dropout is a training-time perturbation)
    sd = lo_e.adapter_state_dict()
    assert set(sd) == {f"base_model.model.{k}.lora_{ab}.weight" for k in lo_o.A for ab in "AB"}
    assert torch.equal(sd["base_model.model.model.layers.1.mlp.up_proj.lora_B.weight"], lo_o.B["model.layers.1.mlp.up_proj"].detach())
    before_lm = {k: m.text[k].clone() for k in ("l0.qkv_w", "l1.down_w", "lm_head", "embed")}
    a0 = lo_e.master[(0, "qkv", "A")].clone()
    gn = tr.updater.optimizer_step(lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    m.engine.sync()
    assert gn.item() > 0 and not torch.equal(lo_e.master[(0, "qkv", "A")], a0) and float(lo_e.arena.abs().max()) == 0.0
    assert all(torch.equal(m.text[k], v) for k, v in before_lm.items())
    assert float((lo_e.master[(0, "qkv", "B")] * (1 - lo_e.mask[(0, "qkv", "B")])).abs().max()) == 0.0      # off-block entries stay zero


def test_lora_with_zero_b_is_the_base_model_bit_for_bit_and_merge_matches_the_adapted_forward(tmp_path):
    from video_r2_amd.lora import LoraTrainer, merged_state_dict
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.train import Trainer
    ocfg = om.cfg_tiny(2)
    w = om.make_weights(ocfg, seed=78)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    r = _rows(m.cfg, 78)[0]
    from video_r2_amd.sft import split_supervised
    prompt, comp, mask = split_supervised(r["input_ids"], r["labels"])
    eng = m.engine
    (emb, plan), = eng.encode_videos([dict(frames_u8=r["videos"][0][0], resize_to=None)], per_prompt=[1])
    emb = emb.clone()
    comp_d = torch.from_numpy(comp[None]).cuda()
    vis_tr = None
    base = Trainer(eng, m.text)
    lp_base = base.forward(prompt, emb, plan, [1.0], comp_d, 1.0).clone()
    del base
    lo = LoraTrainer(eng, m.text, rank=8, alpha=16, dropout=0.1, seed=9)          # fresh adapters: B = 0
    tr = Trainer(eng, m.text, train_lm=False, lora=lo)
    lp0 = tr.forward(prompt, emb, plan, [1.0], comp_d, 1.0).clone()
    eng.sync()
    assert torch.equal(lp0, lp_base)                                               # exact zeros joined in fp32: the base model, bit for bit
    # non-zero B, no dropout: the adapted forward == the forward of the merged weights (up to the merge's bf16 rounding of W + s B A)
    lo_o, lo2 = _lora_pair(ocfg, m, w, 8, 16, 0.0, 0.02)
    tr2 = Trainer(eng, m.text, train_lm=False, lora=lo2)
    lp_ad = tr2.forward(prompt, emb, plan, [1.0], comp_d, 1.0).clone()
    lo2.save_adapter(str(tmp_path / "adapter"), base_model_name_or_path="base")
    from safetensors.torch import load_file
    import json
    cfg_json = json.loads((tmp_path / "adapter" / "adapter_config.json").read_text())
    assert cfg_json["r"] == 8 and cfg_json["lora_alpha"] == 16 and cfg_json["peft_type"] == "LORA" and len(cfg_json["target_modules"]) == 14
    merged = merged_state_dict(w, load_file(str(tmp_path / "adapter" / "adapter_model.safetensors")), cfg_json["lora_alpha"] / cfg_json["r"])
    ref = lo_o.merged(w)
    for k in ("model.layers.0.self_attn.k_proj.weight", "model.layers.1.mlp.gate_proj.weight"):
        assert torch.allclose(merged[k].float(), ref[k], atol=1e-6)
    m2 = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), merged, device="cuda:0")
    (emb2, plan2), = m2.engine.encode_videos([dict(frames_u8=r["videos"][0][0], resize_to=None)], per_prompt=[1])
    lp_m = Trainer(m2.engine, m2.text).forward(prompt, emb2.clone(), plan2, [1.0], comp_d, 1.0).clone()
    m2.engine.sync()
    d = rel_err(lp_ad.cpu().numpy(), lp_m.cpu().numpy())
    assert d.max() < 2e-2 and np.median(d) < 3e-3 and not torch.equal(lp_ad, lp_base)


def _sft_dp_worker(rank, world, port, q, mode):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      VR2_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from video_r2_amd import dist as vdist
    from video_r2_amd.lora import LoraTrainer
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    torch.cuda.set_device(0)
    vdist.init(device="cuda:0")
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=31), device="cuda:0")
    lora_flags = ["--freeze_llm", "True", "--lora_enable", "True", "--lora_rank", "8", "--lora_alpha", "16", "--lora_namespan_exclude", "['lm_head', 'embed_tokens']"]
    args = params.parse_sft_args(["--freeze_vision_tower", "True", "--freeze_merger", "True", "--learning_rate", "1e-3", "--warmup_steps", "0",
                                  "--gradient_accumulation_steps", "2", "--max_steps", "2", "--grad_exchange_dtype", "bf16" if mode == "lora" else "fp32"]
                                 + (lora_flags if mode == "lora" else []))
    lora = LoraTrainer(m.engine, m.text, rank=8, alpha=16, dropout=0.05, seed=args.seed) if mode == "lora" else None
    pool = _rows(m.cfg, 31, n_text=7)                        # 8 rows: rank r takes batches r, r + 2, ... of the permutation
    tr = QwenSFTTrainer(m, args, pool, lambda ex: ex, lora=lora)
    steps = list(tr.iter_steps())
    out = tr.step(steps[0])
    out = tr.step(steps[1])
    torch.cuda.synchronize()
    assert tr.updater._works == {}
    sig = (float(lora.master[(1, "gu", "B")].double().sum()) + float(lora.master[(0, "qkv", "A")].double().sum())) if lora is not None \
        else float(m.text["l0.o_w"].float().sum()) + float(m.text["l1.ln2"].sum())
    q.put((rank, [len(mb[0]["input_ids"]) for mb in steps[0]], out["metrics"]["loss"], out["metrics"]["grad_norm"], sig))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["full", "lora"])
def test_two_rank_sft_steps_keep_replicas_identical(mode):
    """world_size 2 on the one GPU (gloo): the ranks take different rows (batch k of the epoch's permutation goes to rank k mod 2),
    average their gradients -- the LM's buckets, or the adapters' single bucket on the bf16 wire -- and end two optimiser steps with
    bit-identical weights / adapters and the same reduced metrics"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sft_dp_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, rows0, l0, g0, s0), (r1, rows1, l1, g1, s1) = res
    assert len(rows0) == len(rows1) == 2                       # two accumulated micro-batches per rank and step
    assert l0 == l1 and g0 == g1 and g0 > 0 and s0 == s1


def test_num_lora_modules_keeps_the_last_modules_in_module_order():
    """--num_lora_modules 3 (find_target_linear_names, train_sft.py:24-39: `lora_module_names[-num_lora_modules:]`): the last decoder
    layer's gate / up / down projections and nothing else"""
    from video_r2_amd.lora import LoraTrainer
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=12), device="cuda:0")
    lo = LoraTrainer(m.engine, m.text, rank=8, alpha=16, dropout=0.0, num_modules=3, seed=1)
    L, t = ocfg.text.num_hidden_layers, ocfg.text
    assert lo.module_names() == [f"model.layers.{L - 1}.mlp.{n}_proj" for n in ("down", "gate", "up")]
    assert osft.target_module_names(L, 3) == [f"model.layers.{L - 1}.mlp.{n}_proj" for n in ("gate", "up", "down")]
    assert lo.n_params() == 8 * (2 * (t.hidden_size + t.intermediate_size) + (t.intermediate_size + t.hidden_size))
    assert set(k.split(".lora_")[0] for k in lo.adapter_state_dict()) == {f"base_model.model.model.layers.{L - 1}.mlp.{n}_proj" for n in ("gate", "up", "down")}
    for key, mk in lo.mask.items():
        on = key[0] == L - 1 and key[1] in ("gu", "down")
        assert (float(mk.sum()) > 0) == on, key
        assert float(lo.master[key].abs().sum()) == 0.0 or on


def test_lora_beside_a_trainable_merger():
    """--lora_enable True --freeze_merger False (train_sft.py:171-184 of the reference re-enables the merger's requires_grad after
    get_peft_model): adapters on the frozen LM AND the merger's own tensors train in one step; both sets of gradients against autograd,
    and non_lora_state_dict.bin carries the merger (and only the merger) under peft's names"""
    from tests.test_vision_train_gpu import _vision_grads_as_hf
    from video_r2_amd.lora import SITES
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    ocfg = om.cfg_tiny(2)
    w = om.make_weights(ocfg, seed=91)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    args = params.parse_sft_args(["--freeze_llm", "True", "--freeze_vision_tower", "True", "--freeze_merger", "False", "--lora_enable", "True", "--lora_rank", "8",
                                  "--lora_alpha", "16", "--lora_dropout", "0.0", "--lora_namespan_exclude", "['lm_head', 'embed_tokens']", "--merger_lr", "1e-4"])
    lo_o, lo_e = _lora_pair(ocfg, m, w, 8, 16, 0.0, 0.02)
    tr = QwenSFTTrainer(m, args, list(range(4)), lambda ex: ex, lora=lo_e)
    assert tr.updater.vis is not None and tr.updater.vis.train_merger and not tr.updater.vis.train_tower and not tr.updater.train_lm
    rows = _rows(m.cfg, 91, n_text=0)
    n_mb = int((rows[0]["labels"][1:] != -100).sum())
    nll, _ = tr._row(rows[0], 1.0 / n_mb, last=True)
    m.engine.sync()
    o_loss, o_grads = _oracle_step(ocfg, w, rows, lambda k: k.startswith("visual.merger."), lora=lambda r: lo_o.hooks())
    assert abs(float(nll.item()) / n_mb - o_loss) < 2e-3 * abs(o_loss)
    got = _vision_grads_as_hf(tr.updater.vis.state, m.cfg)
    assert set(got) == set(o_grads) and len(got) == 5
    lo_e.finish_grads()
    worst = {k: (got[k] - g).norm().item() / max(g.norm().item(), 1e-12) for k, g in o_grads.items()}
    for l in range(ocfg.text.num_hidden_layers):
        for site in SITES:
            for j, (mod, fin, rws) in enumerate(lo_e.layout[site]):
                name = f"model.layers.{l}.{mod}"
                worst[name + ".A"] = (lo_e.grad[(l, site, "A")][j * 64:j * 64 + 8, :fin].cpu() - lo_o.A[name].grad).norm().item() / lo_o.A[name].grad.norm().item()
                worst[name + ".B"] = (lo_e.grad[(l, site, "B")][rws][:, j * 64:j * 64 + 8].cpu() - lo_o.B[name].grad).norm().item() / lo_o.B[name].grad.norm().item()
    assert max(worst.values()) < 3e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    extra = tr.non_lora_trainables()
    assert set(extra) == {"base_model.model." + k for k in o_grads}


def test_sft_row_against_the_hf_fixture(golden_dir):
    """tests/golden/sft_tiny_g2.npz: HF transformers' loss for `labels` and HF autograd's gradients (oracle/make_golden.py case_sft) on a
    video row with two supervised answer spans -- the engine's SFT row (ViT + merger trainable too) against THEM, not against this
    repo's restatement: loss within 1e-3, the fixture's gradient tensors within 3e-2 of their norm (bf16 GEMM operands)."""
    import os
    from tests.test_vision_train_gpu import _vision_grads_as_hf
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    d = np.load(os.path.join(golden_dir, "sft_tiny_g2.npz"))
    ocfg = om.cfg_tiny(int(d["group"]))
    w = om.make_weights(ocfg, seed=int(d["seed"]))
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    args = params.parse_sft_args(["--freeze_vision_tower", "False", "--freeze_merger", "False", "--freeze_llm", "False"])
    tr = QwenSFTTrainer(m, args, list(range(4)), lambda ex: ex)
    row = dict(input_ids=d["input_ids"], labels=d["labels"], videos=[(torch.from_numpy(d["frames_u8"]), None)], second_per_grid_ts=[float(d["sec"])])
    n = int((d["labels"][1:] != -100).sum())
    nll, cnt = tr._row(row, 1.0 / n, last=True)
    m.engine.sync()
    assert cnt == n and abs(float(nll.item()) / n - float(d["hf_loss"])) < 1e-3 * float(d["hf_loss"])
    got = dict(_engine_grads_as_hf(tr.updater, m.cfg))
    got.update(_vision_grads_as_hf(tr.updater.vis.state, m.cfg))
    worst = {}
    for key in d.files:
        if key.startswith("hf_grad:"):
            name, ref = key[len("hf_grad:"):], torch.from_numpy(d[key])
            g = got[name][:16] if ref.dim() == 2 else got[name]
            worst[name] = ((g - ref).norm() / ref.norm()).item()
    ref = torch.from_numpy(d["hf_lm_head_grad_rows"])
    worst["lm_head.weight[:64]"] = ((got["lm_head.weight"][:64] - ref).norm() / ref.norm()).item()
    print("sft vs HF fixture: loss %.5f vs %.5f; gradient rel err %s" % (float(nll.item()) / n, float(d["hf_loss"]), ", ".join("%s %.2e" % kv for kv in sorted(worst.items()))))
    assert max(worst.values()) < 3e-2, worst
