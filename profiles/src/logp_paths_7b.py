"""7B, one prompt group at the bench shape: per-token log-probs of the same completions from the three paths that produce them --
Engine.score against the prompt's prefix cache (reference scoring), Trainer.forward (the policy's training forward, saved
activations), and the decode's own logits through a teacher-forced Engine.decode -- with the SAME weights: how far apart they are."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from bench import model_cfg, synth_prompt
from video_r2_amd import lib, weights as W
from video_r2_amd.engine import Engine
from video_r2_amd.train import Trainer
lib.load()
device = "cuda:0"
cfg = model_cfg("7b")
sd = W.random_state_dict(cfg, seed=0, device=device)
vis = W.build_vision(cfg, sd, device); pol = W.build_text(cfg, sd, device)
del sd
eng = Engine(cfg, vis, pol, device=device)
upd = Trainer(eng, pol)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
px, grid, ids = synth_prompt(cfg, 32, 0, device)
emb, plan = eng.encode_video(px, [grid])
emb = emb.clone()
st = eng.prefill(pol, ids, emb, plan, [1.0])
toks, _ = eng.decode(pol, st, 8, C, temperature=0.9, top_k=50, seed=1, eos_token_id=None)
toks = torch.clamp(toks, max=cfg.image_token_id - 1)
a = eng.score(pol, st, toks, 0.9).clone()
b = upd.forward(ids, emb, plan, [1.0], toks, 0.9).clone()
eng.sync()
d = (a - b).abs()
print(f"C={C}: score mean logp {a.mean().item():.4f} (min {a.min().item():.3f}); forward mean logp {b.mean().item():.4f}; "
      f"|score - forward| max {d.max().item():.3e} mean {d.mean().item():.3e}; identical positions {(d == 0).float().mean().item():.3f}", flush=True)
