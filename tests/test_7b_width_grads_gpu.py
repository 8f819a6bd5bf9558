"""GPU: the update half at REAL Qwen2.5-VL-7B widths -- hidden 3584, 28 q / 4 kv heads of 128, MLP 18944, vocabulary 152064 -- at
truncated depth (2 LM layers, 1 frozen ViT block) so that torch autograd through the fp32 CPU oracle fits the host: every weight
gradient of the hand-written backward (dgrad GEMMs on the transposed packed copies, the 256-tile weight-gradient GEMM with its
k tail, flash-style attention backward with a GQA group of 7, SwiGLU / RMSNorm / rotary backward, the 152k-wide log-softmax
gradient, embedding rows) against autograd, tensor by tensor.  tests/test_train_gpu.py holds the same comparison on the tiny
configs; this case puts the real tile tails (18944 = 148 x 128, 37888-row gate/up, 152064) and T = 328 token rows (five 64-row
k-tiles + a tail of 8) under it.  Slow (~2-3 min: 1.6 B parameters of autograd on the host)."""
import numpy as np
import pytest
import torch

from oracle import grpo as og
from oracle import ints
from oracle import model as om
from tests.helpers import pixels_from_frames, rel_err, to_engine_cfg
from tests.test_train_gpu import _engine_grads_as_hf, _oracle_grads

pytestmark = pytest.mark.gpu


def test_7b_widths_weight_gradients_vs_autograd():
    from video_r2_amd import weights as W
    from video_r2_amd.engine import Engine
    from video_r2_amd.host import grpo as hg
    from video_r2_amd.host.processor import video_prompt_ids
    from video_r2_amd.train import Trainer
    torch.set_num_threads(min(32, torch.get_num_threads()))
    full = om.cfg_7b()
    ocfg = om.Cfg(vision=om.VisionCfg(**{**full.vision.__dict__, "depth": 1, "fullatt_block_indexes": (0,)}),
                  text=om.TextCfg(**{**full.text.__dict__, "num_hidden_layers": 2}))
    w = om.make_weights(ocfg, seed=31)
    cfg = to_engine_cfg(ocfg)
    dev = "cuda:0"
    eng = Engine(cfg, W.build_vision(cfg, w, dev), W.build_text(cfg, w, dev), device=dev)
    g = torch.Generator().manual_seed(33)
    frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8).numpy()
    px, grid = pixels_from_frames(frames, cfg.vision)
    txt = torch.randint(1000, 150000, (34,), generator=g).tolist()
    prompt = np.asarray(video_prompt_ids(cfg, grid[0] * grid[1] * grid[2] // 4, txt[:8], txt[8:]))      # 6 video tokens + 36
    G, C = 3, 96
    comp = torch.randint(1000, 150000, (G, C), generator=g)
    comp[1, 70] = cfg.eos_token_id                                     # one sequence ends early: masked tail
    tr = Trainer(eng, eng.text)
    emb, plan = eng.encode_video(torch.from_numpy(px).to(dev), [grid])
    emb = emb.clone()
    lp = tr.forward(prompt, emb, plan, [1.0], comp, 0.9)
    mask, _, _ = ints.completion_mask(comp.numpy(), cfg.eos_token_id)
    adv = og.advantages(np.asarray([[1.0], [0.0], [0.5]], dtype=np.float32), [1.0], G)["advantages"]
    with torch.no_grad():
        ref_lp = lp.cpu() + 0.05 * torch.randn(G, C, generator=torch.Generator().manual_seed(1))
    out = hg.grpo_loss(lp, torch.from_numpy(mask).to(dev), torch.from_numpy(adv).to(dev), ref_logps=ref_lp.to(dev), beta=0.04,
                       loss_type="bnpo", stream=torch.cuda.current_stream().cuda_stream)
    tr.backward(out["dlogp"])
    eng.sync()
    o_lp, o_loss, o_grads = _oracle_grads(ocfg, w, px, grid, prompt, comp, adv, mask, ref_lp, "bnpo")
    r = rel_err(lp.cpu().numpy(), o_lp.numpy())[mask.astype(bool)]
    print("7B widths, 2 layers, training forward: logp rel err mean %.2e p95 %.2e max %.2e; loss %.6f vs %.6f"
          % (r.mean(), np.quantile(r, 0.95), r.max(), out["loss"].item(), o_loss))
    assert r.mean() < 1e-3 and r.max() < 4e-3
    assert abs(out["loss"].item() - o_loss) <= 2e-3 * max(abs(o_loss), 1e-2)
    got = _engine_grads_as_hf(tr, cfg)
    assert set(got) == set(o_grads)
    worst = {}
    for k, gref in o_grads.items():
        worst[k] = (got[k] - gref).norm().item() / max(gref.norm().item(), 1e-12)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("7B widths, 2 layers: weight gradients vs autograd, max rel err %.3e (median %.3e); worst: %s"
          % (max(worst.values()), float(np.median(list(worst.values()))), ", ".join("%s %.2e" % kv for kv in top)))
    # bf16 operands in every backward GEMM (dY and X rounded to 8 bits): 3e-2 of each tensor's gradient norm, as on the tiny configs
    for k, e in worst.items():
        assert e < 3e-2, (k, e)
    ge = got["model.embed_tokens.weight"]
    assert ge[cfg.video_token_id].abs().max().item() == 0             # frozen vision features: no embedding gradient at their rows
