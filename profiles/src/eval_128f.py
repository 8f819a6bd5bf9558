"""Evaluation-side throughput at the reference's eval recipe: `generate_until` of video_r2_amd/eval_adapter.py over
the 7B engine with the harness arguments of /root/reference/eval/scripts/accuracy/run_all.sh:31 -- max_pixels=151200,
min_pixels=100352, max_num_frames=128 (FPS_MAX_FRAMES=128) -- on synthetic 64-second clips (128 frames at 2 fps, 360 x 640 source
-> 280 x 504 per frame: 64 x 10 x 18 = 11,520 video tokens per prompt), random-init weights, greedy decoding (temperature 0, the
harness default), `--new` new tokens per request (random-init weights never emit EOS, so every request generates all of them).
The reference answers ONE request at a time (--batch_size 1); the adapter answers `--batch` requests per vision pass / prefill /
decode.  Prints one JSON line.    python profiles/src/eval_128f.py [--batch 8] [--requests 8] [--new 256] [--frames 128]"""
import argparse, json, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, '.')
os.environ["FPS_MAX_FRAMES"] = os.environ.get("FPS_MAX_FRAMES", "128")
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--requests", type=int, default=8)
ap.add_argument("--new", type=int, default=256); ap.add_argument("--frames", type=int, default=128)
a = ap.parse_args()
os.environ["FPS_MAX_FRAMES"] = str(a.frames)
from video_r2_amd import lib, weights as W
from video_r2_amd.config import ModelConfig
from video_r2_amd.eval_adapter import VideoR2EvalModel
from video_r2_amd.host.frontend import ByteTokenizer
from video_r2_amd.model import VideoR2Model
lib.load()
dev = "cuda:0"
cfg = ModelConfig.qwen25vl_7b()
sd = W.random_state_dict(cfg, seed=0, device=dev)
m = VideoR2Model(cfg, W.build_vision(cfg, sd, dev), W.build_text(cfg, sd, dev), device=dev)
del sd
torch.cuda.empty_cache()
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(0)
docs = []
for i in range(a.requests + a.batch):          # + one warm-up batch
    p = os.path.join(tmp, f"clip{i}.npz")
    np.savez(p, frames=rng.integers(0, 256, size=(a.frames, 360, 640, 3), dtype=np.uint8), fps=np.float32(2.0))
    docs.append({"video": p, "q": f"Question {i}: what happens in the clip, and in which order?"})
task = {"videomme_think": {"test": docs}}
ev = VideoR2EvalModel(model=m, tokenizer=ByteTokenizer(cfg), task_dict=task, max_num_frames=a.frames, min_pixels=100352, max_pixels=151200,
                      batch_size=a.batch)


class Req:
    def __init__(self, *args):
        self.args = args


kw = {"max_new_tokens": a.new, "temperature": 0, "until": ["\n\n"]}
mk = lambda i: Req(docs[i]["q"], kw, (lambda d: [d["video"]]), i, "videomme_think", "test")
ev.generate_until([mk(a.requests + j) for j in range(a.batch)])           # warm-up: code objects, workspaces, the decode graph
torch.cuda.synchronize()
ids0, clips0, _, _ = ev._prepare(docs[0]["q"], [docs[0]["video"]])
t0 = time.perf_counter()
out = ev.generate_until([mk(i) for i in range(a.requests)])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"what": "lmms-eval generate_until over the MI355X engine, reference eval recipe (run_all.sh:31)", "model": "Qwen2.5-VL-7B random-init bf16",
                  "frames_per_request": a.frames, "frame_hw": list(clips0[0]["resize_to"]), "prompt_tokens": int(len(ids0)),
                  "requests": a.requests, "batch_size": a.batch, "new_tokens_per_request": a.new, "wall_s": dt,
                  "requests_per_s": a.requests / dt, "new_tokens_per_s": a.requests * a.new / dt, "s_per_request": dt / a.requests,
                  "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                  "note": "video read (npz), frame pick, device resize + patchify, ViT, prefill, greedy decode, detokenise: all inside wall_s"}))
