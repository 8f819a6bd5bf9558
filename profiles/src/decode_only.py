"""Decode-only run at 7B for profiling: P prompt groups x G=8 rows, C new tokens (graph replay)."""
import os, sys, time, torch
sys.path.insert(0, '.')
from bench import model_cfg, synth_prompt
from video_r2_amd import lib, weights as W
from video_r2_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MODE = sys.argv[3] if len(sys.argv) > 3 else "bf16"          # bf16 | fp8 (weights + activations) | fp8w (weight-only)
lib.load()
device = "cuda:0"
cfg = model_cfg("7b")
sd = W.random_state_dict(cfg, seed=0, device=device)
vis = W.build_vision(cfg, sd, device); pol = W.build_text(cfg, sd, device)
del sd
eng = Engine(cfg, vis, pol, device=device)
if MODE != "bf16":
    W.add_fp8_decode_copies(pol)
    eng.decode_fp8, eng.decode_fp8_act = True, MODE == "fp8"
prompts = [synth_prompt(cfg, 32, i, device) for i in range(P)]
slots = eng.prefix_slots(P, max(len(p[2]) for p in prompts), tag="pol")
states = []
for p_, (px, grid, ids) in enumerate(prompts):
    emb, plan = eng.encode_video(px, [grid])
    states.append(eng.prefill(pol, ids, emb.clone(), plan, [1.0], tag="pol", prefix_out=slots[p_]))
def run(tag=""):
    for rep in range(1 if os.environ.get('VR2_NO_GRAPH') else 2):
        eng.sync(); t0 = time.perf_counter()
        toks, _ = eng.decode(pol, states, 8, C, temperature=0.9, top_k=50, seed=rep, eos_token_id=None, use_graph=not os.environ.get('VR2_NO_GRAPH'))
        eng.sync(); dt = time.perf_counter() - t0
        print(f"P={P} rows={8*P} C={C} {MODE}{tag}: {dt*1e3/C:.3f} ms/token", flush=True)
    return toks


# VR2_AB="3=0;3=1": launch-time tuning knobs (vr2_tune_set key=value[,key=value]) A/B-ed in ONE process, interleaved twice; the
# captured graphs are dropped between settings (a graph keeps the kernel variant it was captured with)
ab = [v for v in os.environ.get("VR2_AB", "").split(";") if v]
if not ab:
    run()
else:
    ref = None
    for rnd in range(2):
        for setting in ab:
            for kv in setting.split(","):
                k, v = kv.split("=")
                lib.load().vr2_tune_set(int(k), int(v))
            for ex_ in eng._graphs.values():
                lib.call("vr2_graph_destroy", ex_)
            eng._graphs.clear()
            toks = run(f" [{setting}]").cpu()
            if ref is None:
                ref = toks
            print(f"   tokens equal to the first setting's: {bool((toks == ref).all())}", flush=True)
