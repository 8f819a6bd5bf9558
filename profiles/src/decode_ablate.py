"""Upper bounds for launch fusion in the decode step: the same decode (7B, P groups x 8 rows, C tokens, graph replay) with named
entry points NOT launched (results wrong by construction; the graph is captured without them, so their kernels, their launch
boundaries and their slab traffic are all gone).  What a perfect fusion of that kernel into its neighbours could at most save.
Usage: python profiles/src/decode_ablate.py P C     (settings run in one process, the full step first and last)"""
import sys, time, torch
sys.path.insert(0, '.')
from bench import model_cfg, synth_prompt
from video_r2_amd import lib, weights as W
from video_r2_amd.engine import Engine
import video_r2_amd.engine as E
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib.load()
device = "cuda:0"
cfg = model_cfg("7b")
sd = W.random_state_dict(cfg, seed=0, device=device)
vis = W.build_vision(cfg, sd, device); pol = W.build_text(cfg, sd, device)
del sd
eng = Engine(cfg, vis, pol, device=device)
prompts = [synth_prompt(cfg, 32, i, device) for i in range(P)]
slots = eng.prefix_slots(P, max(len(p[2]) for p in prompts), tag="pol")
states = []
for p_, (px, grid, ids) in enumerate(prompts):
    emb, plan = eng.encode_video(px, [grid])
    states.append(eng.prefill(pol, ids, emb.clone(), plan, [1.0], tag="pol", prefix_out=slots[p_]))
real_call = lib.call
skip = set()


def call(name, *a):
    if name in skip:
        return 0
    return real_call(name, *a)


E.lib.call = call
settings = [("full step", []), ("without rope_append_decode", ["vr2_rope_append_decode"]),
            ("without the two add_rmsnorm launches per layer", ["vr2_add_rmsnorm"]),
            ("without attention (+ combine)", ["vr2_attn_decode"]),
            ("without rope, norms and attention", ["vr2_rope_append_decode", "vr2_add_rmsnorm", "vr2_attn_decode"]),
            ("without the q/k/v and o GEMVs' launches (gate/up, down, lm_head kept)", ["__qkv_o__"]),
            ("full step again", [])]
real_gemv = eng._gemv


def gemv(x, Pm, wname, G, out, nsplit, prequant=None):
    if "__qkv_o__" in skip and (wname.endswith("qkv_w") or wname.endswith("o_w")):
        return
    return real_gemv(x, Pm, wname, G, out, nsplit, prequant=prequant)


eng._gemv = gemv
for tag, names in settings:
    skip.clear(); skip.update(names)
    for ex_ in eng._graphs.values():
        real_call("vr2_graph_destroy", ex_)
    eng._graphs.clear()
    best = None
    for rep in range(2):
        eng.sync(); t0 = time.perf_counter()
        eng.decode(pol, states, 8, C, temperature=0.9, top_k=50, seed=rep, eos_token_id=None)
        eng.sync(); dt = (time.perf_counter() - t0) * 1e3 / C
        best = dt if best is None else min(best, dt)
    print(f"P={P} rows={8 * P} C={C}: {best:.3f} ms/token  {tag}", flush=True)
