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
    return float(total), {k: v.grad for k, v in wt.items() if v.requires_grad and v.grad is not None}


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
    assert mult == 1.0 and abs(o1["metrics"]["learning_rate"] - 1e-3) < 1e-12 and o1["metrics"]["grad_norm"] > 0
    st, vs = tr.updater.state, tr.updater.vis.state

    def moved(cur, old):
        d = (cur - old).abs()
        return float(d[d > 0].median())
    # Adam's second step after a zero-lr first step: m / sqrt(v) of two equal gradients is 1 up to eps -> |delta| = lr of the group
    assert abs(moved(st.master["l0.o_w"], before["l0.o_w"]) - 1e-3) < 2e-5
    assert abs(moved(vs.master["b0.qkv_w"], before["b0.qkv_w"]) - 2e-4) < 4e-6
    assert abs(moved(vs.master["fc2_w"], before["fc2_w"]) - 5e-4) < 1e-5
    assert tr.global_step == 2 and tr.num_input_tokens_seen == 2 * (len(rows[0]["input_ids"]) + len(rows[1]["input_ids"]))
