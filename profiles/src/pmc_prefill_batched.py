"""ViT encode (16 clips x 32 frames in ONE pass) + LM prefill of the 16 prompts in ONE row space at 7B, a few passes, for rocprofv3
PMC collection (MFMA utilisation of the vision-prefill GEMMs, round 2: Engine.encode_videos / prefill_batch)."""
import sys
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
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prompts = [synth_prompt(cfg, 32, i, device) for i in range(P)]
slots = eng.prefix_slots(P, max(len(p[2]) for p in prompts), tag="pol")
for rep in range(3):
    enc = eng.encode_videos([dict(pixel_values=px, grid=grid) for px, grid, _ in prompts])
    st = eng.prefill_batch(pol, [(ids, enc[i][0], enc[i][1], [1.0]) for i, (_, _, ids) in enumerate(prompts)], slots, tag="pol")
eng.sync()
print("done")
