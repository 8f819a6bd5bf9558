"""ViT encode (32 frames) + LM prefill at 7B, a few passes, for rocprofv3 PMC collection (MFMA utilisation of the vision-prefill GEMMs)."""
import sys, torch
sys.path.insert(0, '.')
from bench import model_cfg, synth_prompt
from video_r2_amd import lib, weights as W
from video_r2_amd.engine import Engine
lib.load()
device = "cuda:0"
cfg = model_cfg("7b")
sd = W.random_state_dict(cfg, seed=0, device=device)
vis = W.build_vision(cfg, sd, device); pol = W.build_text(cfg, sd, device)
del sd
eng = Engine(cfg, vis, pol, device=device)
px, grid, ids = synth_prompt(cfg, 32, 0, device)
for rep in range(3):
    emb, plan = eng.encode_video(px, [grid])
    st = eng.prefill(pol, ids, emb, plan, [1.0])
eng.sync()
print("done")
