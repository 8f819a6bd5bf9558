"""GPU: LoRA at REAL Qwen2.5-VL-7B widths (hidden 3584, 28 q / 4 kv heads of 128, MLP 18944, vocabulary 152064) and the shipped
recipe's adapter shape (rank 64, alpha 64, dropout 0.05 -- train_sft.sh:30-37 of the reference), at truncated depth (2 LM layers,
1 frozen ViT block) so that autograd through the fp32 CPU oracle fits the host: the supervised loss and the gradient of every adapter
tensor (28 of them) against autograd, with the engine's own dropout draws replayed in the oracle.  Puts the real shapes under the
adapter path: A of 64 x 18944 on the SwiGLU output, the gate/up projection with 128 extra K columns (3584 + 128), block-structured
B of 4608 x 192 and 37888 x 128.  tests/test_sft_gpu.py holds the same comparison on the tiny configs (rank 8 padded to 64)."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import sft as osft
from tests.helpers import pixels_from_frames, to_engine_cfg
from tests.test_sft_gpu import _dropout_masks, _lora_pair, _oracle_step
from video_r2_amd import params

pytestmark = pytest.mark.gpu


def test_7b_widths_lora_adapter_gradients_vs_autograd():
    from video_r2_amd.lora import SITES
    from video_r2_amd.model import VideoR2Model
    from video_r2_amd.sft import QwenSFTTrainer
    torch.set_num_threads(min(32, torch.get_num_threads()))
    full = om.cfg_7b()
    ocfg = om.Cfg(vision=om.VisionCfg(**{**full.vision.__dict__, "depth": 1, "fullatt_block_indexes": (0,)}),
                  text=om.TextCfg(**{**full.text.__dict__, "num_hidden_layers": 2}))
    w = om.make_weights(ocfg, seed=35)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), w, device="cuda:0")
    cfg = m.cfg
    p, r, alpha = 0.05, 64, 64
    args = params.parse_sft_args(["--freeze_llm", "True", "--freeze_vision_tower", "True", "--freeze_merger", "True", "--lora_enable", "True",
                                  "--lora_rank", str(r), "--lora_alpha", str(alpha), "--lora_dropout", str(p),
                                  "--lora_namespan_exclude", "['lm_head', 'embed_tokens']"])
    lo_o, lo_e = _lora_pair(ocfg, m, w, r, alpha, p, 0.01)
    assert lo_e.rp == 64 and lo_e.n_params() == 2 * r * 90112          # per layer: sum of (in + out) over q k v o gate up down; x 28 layers = 161.5 M
    g = torch.Generator().manual_seed(36)
    frames = torch.randint(0, 256, (2, 3, 56, 84), generator=g, dtype=torch.uint8)
    px, grid = pixels_from_frames(frames.numpy(), cfg.vision)
    tv = grid[0] * grid[1] * grid[2] // 4
    rnd = lambda n: torch.randint(1000, 150000, (n,), generator=g).tolist()      # noqa: E731
    ids, labels = osft.labels_of([(rnd(8), False), ([cfg.vision_start_token_id] + [cfg.video_token_id] * tv + [cfg.vision_end_token_id] + rnd(20), False),
                                  (rnd(70) + [cfg.eos_token_id], True), (rnd(9), False), (rnd(40) + [cfg.eos_token_id], True)])
    rows = [dict(input_ids=ids, labels=labels, videos=[(frames, None)], second_per_grid_ts=[1.0], _px=px, _grid=grid)]
    tr = QwenSFTTrainer(m, args, list(range(4)), lambda ex: ex, lora=lo_e)
    n_mb = int((labels[1:] != -100).sum())
    masks = _dropout_masks(lo_e, lo_o, rows, p, 2)
    nll, n = tr._row(rows[0], 1.0 / n_mb, last=True)
    m.engine.sync()
    loss = float(nll.item()) / n_mb

    def hooks(row):
        lo_o.set_dropout(masks[0], p)
        return lo_o.hooks()
    o_loss, _ = _oracle_step(ocfg, w, rows, lambda k: False, lora=hooks)
    lo_e.finish_grads()
    torch.cuda.synchronize()
    worst = {}
    for l in range(2):
        for site in SITES:
            gA, gB = lo_e.grad[(l, site, "A")], lo_e.grad[(l, site, "B")]
            for j, (mod, fin, rws) in enumerate(lo_e.layout[site]):
                name = f"model.layers.{l}.{mod}"
                for got, ref, kind in ((gA[j * 64:(j + 1) * 64, :fin].cpu(), lo_o.A[name].grad, "A"), (gB[rws][:, j * 64:(j + 1) * 64].cpu(), lo_o.B[name].grad, "B")):
                    worst[name + "." + kind] = (got - ref).norm().item() / max(ref.norm().item(), 1e-12)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("7B widths, 2 layers, LoRA r=64 p=0.05, T=%d rows: loss %.5f vs %.5f; adapter gradients max rel err %.3e (median %.3e); worst %s"
          % (len(ids), loss, o_loss, max(worst.values()), float(np.median(list(worst.values()))), ", ".join("%s %.2e" % kv for kv in top)))
    assert abs(loss - o_loss) < 2e-3 * abs(o_loss)
    assert len(worst) == 28 and max(worst.values()) < 3e-2, top
