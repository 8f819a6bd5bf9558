"""Full-size rehearsal of the shell entry point: a random-init Qwen2.5-VL-7B checkpoint directory in HF layout, a small dataset of
array clips at the reference geometry (360 x 640 source frames -> 280 x 504 under --video_max_pixels 360*420, <= 32 frames) with one
still-image row, then `bash train/scripts/train_grpo.sh` on ONE GPU for a few optimiser steps.  Not a test (it writes ~17 GB and
runs for minutes): it answers "does the recipe fit and run at 7B through the entry point a user launches".
Usage: python profiles/src/full_size_entry.py <workdir> [steps] [groups_per_step] [max_completion_length] [frames per clip @ 30 fps]
(VR2_ENTRY_FLAGS="--freeze_vision_tower False ..." appends flags to the launch line)"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
work = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 2
C = int(sys.argv[4]) if len(sys.argv) > 4 else 256
NF = int(sys.argv[5]) if len(sys.argv) > 5 else 300
os.makedirs(work, exist_ok=True)
model_dir = os.path.join(work, "model7b")
if not os.path.exists(os.path.join(model_dir, "model.safetensors")):
    from bench import model_cfg
    from video_r2_amd import weights as W
    from safetensors.torch import save_file
    cfg = model_cfg("7b")
    t0 = time.time()
    sd = W.random_state_dict(cfg, seed=0, device="cuda:0")
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump(cfg.to_dict(), f)
    save_file({k: v.to(torch.bfloat16).contiguous().cpu() for k, v in sd.items()}, os.path.join(model_dir, "model.safetensors"), metadata={"format": "pt"})
    del sd
    torch.cuda.empty_cache()
    print(f"[entry7b] checkpoint written in {time.time() - t0:.0f} s", flush=True)
clips = os.path.join(work, "videos")
os.makedirs(os.path.join(clips, "set"), exist_ok=True)
rng = np.random.default_rng(0)
rows = []
n_rows = groups * (steps + 1)
for i in range(n_rows):
    if i == 1:
        from PIL import Image
        Image.fromarray(rng.integers(0, 256, size=(360, 640, 3), dtype=np.uint8)).save(os.path.join(clips, "set", "p.png"))
        rows.append({"image": "set/p.png", "conversations": [{"from": "human", "value": "<image>\nWhat is shown?\nA. x\nB. y"},
                                                             {"from": "gpt", "value": "<answer>A</answer>"}], "temporal_grounding": {}})
        continue
    np.savez(os.path.join(clips, "set", f"c{i}.npz"), frames=rng.integers(0, 256, size=(NF, 360, 640, 3), dtype=np.uint8), fps=np.float32(30.0))
    rows.append({"video": f"set/c{i}.npz", "key_frame": "",
                 "conversations": [{"from": "human", "value": f"<video>\nQuestion {i}?\nA. yes\nB. no"},
                                   {"from": "gpt", "value": "<think>..</think><answer>B</answer>"}],
                 "temporal_grounding": {"00:01-00:03": "something happens."}})
data = os.path.join(work, "grpo.json")
with open(data, "w") as f:
    json.dump(rows, f)
env = dict(os.environ, NUM_DEVICES="1", MODEL_NAME=model_dir, DATA_PATH=data, DATA_FOLDER=clips, SUBTITLES_FOLDER=os.path.join(work, "subs"),
           OUTPUT_DIR=os.path.join(work, "out"), MASTER_ADDR="127.0.0.1")
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
    env.pop(k, None)
flags = ["--per_device_train_batch_size", str(8 * groups), "--max_completion_length", str(C), "--max_steps", str(steps),
         "--reward_func_names", "accuracy_reward,format_reward", "--reward_weights", "1.0,1.0", "--logging_steps", "1",
         "--save_steps", "100000", "--image_max_pixels", str(360 * 420)] + os.environ.get("VR2_ENTRY_FLAGS", "").split()
t0 = time.time()
r = subprocess.run(["bash", os.path.join(ROOT, "train", "scripts", "train_grpo.sh"), *flags], env=env, capture_output=True, text=True)
print(r.stdout[-4000:])
print("---- stderr ----")
print(r.stderr[-5000:])
print(f"[entry7b] exit {r.returncode}, {time.time() - t0:.0f} s wall for {steps} steps of {groups} groups x 8 completions x {C} tokens", flush=True)
sys.exit(r.returncode)
