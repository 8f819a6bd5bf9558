"""GRPO rollout samples/sec on MI355X (BASELINE.json metric) -- one prompt group per GPU per step, weak scaling.

    python bench.py --gpus N --steps K --warmup W

N>1: one rank per GPU over RCCL.  Either the caller starts the ranks (python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or -- when WORLD_SIZE is not set
-- this process starts them itself as a child `python -m torch.distributed.run` (as the reference launches its trainer,
/root/reference/train/scripts/train_grpo.sh:41) and only forwards the one JSON line; the parent never touches the GPU.

A "step" = one optimiser step = one pass of the hot path over P prompt groups per rank (default P = 22: the reference's
per_device_train_batch_size x gradient_accumulation_steps in completions / G; 22 groups of G = 8 = 176 sequences share ONE decode
weight pass -- the register-ring GEMV serves up to 256 rows, and a 176-row token costs 1.19x a 128-row one for 1.375x the
sequences; P = 32 is 9.7 % faster still per sample but makes the driver's 25-step run 1.7x as long), inputs resident in HBM:
  ViT encode (32 frames) -> LM prefill (once) -> shared-prefix decode of G=8 completions x C new tokens (hipGraph) ->
  reference-model log-probs over the unified [prompt | completions] rows -> policy training forward (log-probs, activations saved) -> group
  advantages -> GRPO loss -> backward -> gradient all-reduce (N>1) -> global-norm clip + AdamW.
`value` = completions produced by all ranks / wall time of the K timed steps (max over ranks).
Prints ONE JSON line on rank 0 with `roofline` = the kernel with the largest share of the step's kernel time in the tracked
rocprofv3 summary (profiles/*kernel_stats_default_bench_p22.md: `gemm256_kernel<4>`, the gate/up GEMM + SwiGLU of the training /
scoring passes, MFMA-bound), timed with HIP events on the engine's stream; `roofline_decode*` = the decode phase's weight-streaming
GEMVs (HBM-bound), `roofline_wgrad` = the largest weight-gradient GEMM; and `cpu_baseline` (the CPU oracle = "port", rank 0,
N=1 only: a bounded sample, or the whole step with --cpu-baseline full at config 1's size).  PMC figures (`traffic`, MFMA-busy)
come from profiles/pmc_traffic.json and are printed only while the kernel's source files are unchanged since the measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md)


def pmc_entry(name):
    """profiles/pmc_traffic.json entry `name` (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc
    passes, MFMA-busy share) -- or None when the file is absent or ANY source file of the kernel has changed since the
    measurement (sha256 recorded per entry by profiles/src/pmc_traffic.py): a counter value is evidence for the code it was
    read from, not for its successors."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            e = json.load(f)["entries"].get(name)
        if e is None:
            return None
        for rel, want in e["files"].items():
            with open(os.path.join(ROOT, rel), "rb") as f:
                if hashlib.sha256(f.read()).hexdigest() != want:
                    return None
        return e
    except (OSError, KeyError, ValueError):
        return None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="7b", choices=["7b", "3b", "tiny"])
    p.add_argument("--frames", type=int, default=32)
    p.add_argument("--G", type=int, default=8)
    p.add_argument("--max-new", type=int, default=1024)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"],
                   help="sample: the CPU oracle on a bounded sample of the workload, extrapolated (default); full: the oracle runs the "
                        "WHOLE step of this workload -- ViT, prefill, G x C sampled decode with KV cache, reference and policy "
                        "log-probs, loss -- on the host cores (minutes at BASELINE config 1's size: --model 3b --frames 4 --G 2 "
                        "--max-new 32 --prompts-per-gpu 1; hours at the headline's)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-update", action="store_true", help="rollout + scoring + loss only (no backward / optimiser)")
    p.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8", "fp8w"],
                   help="fp8 (BASELINE config 5): the decode GEMVs stream per-channel e4m3 copies of the LM weights and, above 16 "
                        "rows, per-row e4m3 activations on the fp8 MFMA; fp8w: weight-only (bf16 activations, bf16 MFMA: round 1). "
                        "With fp8 the no-grad LM GEMMs (rollout prefill, reference prefill + log-probs) run on the same copies with fp8 "
                        "activations too (vr2_gemm_a8w8); the training forward / backward stay bf16; the policy's copies are "
                        "re-quantised after every optimiser step (timed)")
    p.add_argument("--grad-exchange", default="fp32", choices=["fp32", "bf16"],
                   help="wire format of the N > 1 gradient all-reduce: fp32 arena as it is (30.5 GB per step at 7B) or rounded to bf16 "
                        "on the way out (15.2 GB, the figure of the reference's bf16 ZeRO-3 exchange)")
    p.add_argument("--no-gemm-fp8", action="store_true",
                   help="with --decode-weights fp8: keep the no-grad LM GEMMs (rollout prefill, reference prefill + log-probs) on the "
                        "bf16 path instead of per-row e4m3 activations x e4m3 weights on the block-scaled fp8 MFMA (A/B switch)")
    p.add_argument("--launch-check", action="store_true",
                   help="rehearse the N-rank launch only: process group, barrier, max-over-ranks timing and the JSON line, no "
                        "engine work (runs without a GPU; tests/test_bench_launch.py)")
    p.add_argument("--no-recipe-shape", action="store_true",
                   help="skip the extra untimed steps at ONE prompt group per GPU (the reference recipe's per-device shape)")
    p.add_argument("--eos-mix", default="uniform:128:1024",
                   help="one extra, separately reported step pair with EOS-terminated completions emulated (random-init weights "
                        "never emit EOS): sequence lengths drawn uniform:LO:HI per sequence (seeded); the decode stops at the "
                        "longest one (rounded up to the engine's 64-step EOS check) and the scoring / training passes run on the "
                        "packed rows.  'off' skips it.  Never part of `value`.")
    p.add_argument("--prompts-per-gpu", type=int, default=None,
                   help="prompt groups per GPU per optimiser step (the reference's per_device_train_batch_size x "
                        "gradient_accumulation_steps); groups are decoded together, P*G <= 256 rows per weight pass (128 with fp8 decode "
                        "weights).  Default: 176 // G up to 32 frames (22 at G = 8; bf16 or fp8 x fp8 decode), else 128 // G; clamped to the row limit (128 for weight-only fp8)")
    p.add_argument("--prompt-pool-gb", type=float, default=9.0,
                   help="HBM for the resident synthetic prompts: distinct prompts for every step while they fit, else a cyclic pool")
    return p.parse_args()


def model_cfg(name):
    from video_r2_amd.config import ModelConfig
    if name == "7b":
        return ModelConfig.qwen25vl_7b()
    if name == "3b":
        return ModelConfig.qwen25vl_3b()
    return ModelConfig.tiny()               # dry runs / multi-rank rehearsals


def synth_prompt(cfg, n_frames, idx, device):
    """seeded uint8 frames fed post-resize (224^2) -> normalise -> patchify; prompt = 32N video tokens + 64."""
    from video_r2_amd.host import processor as hp
    g = torch.Generator().manual_seed(1234 + idx)
    frames = torch.randint(0, 256, (n_frames, 3, 224, 224), generator=g, dtype=torch.uint8)
    px, grid = hp.patchify(hp.normalize_frames(frames), cfg.vision.patch_size, cfg.vision.spatial_merge_size,
                           cfg.vision.temporal_patch_size)
    tv = grid[0] * grid[1] * grid[2] // cfg.vision.spatial_merge_size ** 2
    hi = min(cfg.text.vocab_size, cfg.image_token_id) - 1
    txt = torch.randint(min(1000, hi // 2), hi, (62,), generator=g).tolist()
    ids = hp.video_prompt_ids(cfg, tv, txt[:8], txt[8:])            # 8 + 1 + Tv + 1 + 54 = Tv + 64
    return px.to(device), grid, np.asarray(ids)


def stub_rewards(comp):
    """hash(completion ids) mod 3 / 2, one reward function, no HTTP."""
    h = (comp.astype(np.int64) * np.arange(1, comp.shape[1] + 1)[None, :]).sum(1)
    return ((h % 3) / 2.0).astype(np.float32)[:, None]


def cpu_baseline(cfg_name, n_frames, G, C):
    """The CPU oracle ("port") timed on this host's cores on a BOUNDED sample of the same workload (about 10-30 s of CPU
    work) and extrapolated linearly in layers / frames / tokens / decode steps; the sample is stated in the result."""
    from oracle import model as om
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    oc = {"7b": om.cfg_7b, "3b": om.cfg_3b}.get(cfg_name, lambda: om.cfg_tiny(2))()
    t, v = oc.text, oc.vision
    L_s, D_s = 1, 2
    f_s = min(4, n_frames)                         # frames in the ViT sample
    s_s = min(256, 32 * n_frames + 64)             # prompt tokens in the prefill sample
    vs = min(t.vocab_size, 32768)
    small = om.Cfg(vision=om.VisionCfg(**{**v.__dict__, "depth": D_s, "fullatt_block_indexes": (1,)}),
                   text=om.TextCfg(**{**t.__dict__, "num_hidden_layers": L_s, "vocab_size": vs}),
                   image_token_id=vs - 6, video_token_id=vs - 5, vision_start_token_id=vs - 4, vision_end_token_id=vs - 3,
                   eos_token_id=vs - 2, pad_token_id=vs - 1)
    w = om.make_weights(small, seed=0, bf16_round=False)
    px = torch.randn(128 * f_s, v.in_channels * v.temporal_patch_size * v.patch_size ** 2)
    grid = np.asarray([[f_s // 2, 16, 16]])
    S = 32 * n_frames + 64
    t0 = time.perf_counter()
    om.vision_forward(w, small, px, grid)
    t_vit = (time.perf_counter() - t0) * (v.depth / D_s) * (n_frames / f_s)
    e = torch.randn(1, s_s, t.hidden_size)
    pos = torch.arange(s_s)[None, None].expand(3, 1, s_s)
    cache = [None] * L_s
    t0 = time.perf_counter()
    om.lm_forward(w, small, e, pos, None, cache)
    t_prefill = (time.perf_counter() - t0) * t.num_hidden_layers / L_s * (S / s_s)
    cache = [(k.expand(G, -1, -1, -1).contiguous(), vv.expand(G, -1, -1, -1).contiguous()) for k, vv in cache]
    x1 = torch.randn(G, 1, t.hidden_size)
    n_dec = 2
    t0 = time.perf_counter()
    for i in range(n_dec):
        om.lm_forward(w, small, x1, torch.full((3, G, 1), s_s + i), None, cache)
    t_dec_step = (time.perf_counter() - t0) / n_dec * t.num_hidden_layers / L_s
    head = torch.randn(vs, t.hidden_size)
    t0 = time.perf_counter()
    (torch.randn(G, t.hidden_size) @ head.t()).softmax(-1)
    t_head = (time.perf_counter() - t0) * t.vocab_size / vs
    per_tok = t_prefill / S
    t_score = 2 * G * C * (per_tok + t_head / G) + t_prefill     # policy + reference scoring, + reference prompt prefill
    total = t_vit + t_prefill + (C - 1) * (t_dec_step + t_head) + t_score
    return {"value": G / total, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"CPU oracle (torch fp32, {threads} threads): {D_s}/{v.depth} ViT blocks on {f_s} frames, {L_s}/{t.num_hidden_layers} LM "
                      f"layers on {s_s} prompt tokens, {n_dec} decode steps (G={G}), lm_head on {vs}/{t.vocab_size} vocab rows; extrapolated "
                      f"linearly to {n_frames} frames, S={S}, G={G}, C={C}: vit {t_vit:.1f}s prefill {t_prefill:.1f}s "
                      f"decode-step {t_dec_step + t_head:.2f}s scoring {t_score:.1f}s"}


def cpu_baseline_full(cfg_name, n_frames, G, C, P):
    """The CPU oracle ("port") running ONE WHOLE step of this workload on the host cores, nothing sampled or extrapolated:
    per prompt group the vision tower, the prompt prefill, G sampled completions of C tokens with a KV cache (oracle generate:
    HF's _sample loop, temperature 0.9 / top-k 50 / multinomial), the reference model's and the policy's per-token log-probs
    (two full forward passes over [prompt | completion], as grpo_trainer.py:826 and :1094 run them) and the GRPO loss.  fp32,
    torch CPU threads; backward and optimiser are not part of it (the oracle has none), so the figure flatters the CPU."""
    from oracle import grpo as og
    from oracle import ints
    from oracle import model as om
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    oc = {"7b": om.cfg_7b, "3b": om.cfg_3b}.get(cfg_name, lambda: om.cfg_tiny(2))()
    t0 = time.perf_counter()
    w = om.make_weights(oc, seed=0)
    t_init = time.perf_counter() - t0
    from video_r2_amd.config import ModelConfig
    ecfg = {"7b": ModelConfig.qwen25vl_7b, "3b": ModelConfig.qwen25vl_3b}.get(cfg_name, ModelConfig.tiny)()
    phases = {"generate": 0.0, "ref_logps": 0.0, "policy_logps": 0.0, "loss": 0.0}
    gen = torch.Generator().manual_seed(0)
    t_all = time.perf_counter()
    with torch.no_grad():
        for i in range(P):
            px, grid, ids = synth_prompt(ecfg, n_frames, i, "cpu")
            ids_t = torch.from_numpy(ids)[None]
            t0 = time.perf_counter()
            out = om.generate(w, oc, ids_t, None, px, np.asarray([grid]), [1.0], C, temperature=0.9, top_k=50,
                              num_return_sequences=G, generator=gen, eos_token_id=None)
            phases["generate"] += time.perf_counter() - t0
            full = torch.clamp(out, max=oc.image_token_id - 1)
            full[:, :ids_t.shape[1]] = ids_t
            mm = dict(pixel_values_videos=px.repeat(G, 1), video_grid_thw=np.asarray([grid] * G), second_per_grid_ts=[1.0] * G)
            t0 = time.perf_counter()
            lp_ref = om.per_token_logps(w, oc, full, np.ones_like(full.numpy()), C, 0.9, **mm)
            phases["ref_logps"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            lp = om.per_token_logps(w, oc, full, np.ones_like(full.numpy()), C, 0.9, **mm)
            phases["policy_logps"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            comp = full[:, ids_t.shape[1]:].numpy()
            mask, _, _ = ints.completion_mask(comp, -1)
            adv = og.advantages(stub_rewards(comp), [1.0], G)["advantages"]
            og.loss(lp.numpy(), mask, adv, ref_per_token_logps=lp_ref.numpy(), beta=0.04, loss_type="bnpo")
            phases["loss"] += time.perf_counter() - t0
    total = time.perf_counter() - t_all
    return {"value": G * P / total, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"in full: the CPU oracle (torch fp32, {threads} threads) ran the WHOLE step -- {P} prompt group(s) x (ViT on {n_frames} "
                      f"frames, prefill, G={G} x {C} sampled tokens with KV cache, reference + policy log-probs, loss) -- in {total:.1f} s "
                      f"(generate {phases['generate']:.1f} s, reference log-probs {phases['ref_logps']:.1f} s, policy log-probs "
                      f"{phases['policy_logps']:.1f} s; weight init {t_init:.1f} s not counted); no backward / optimiser on the CPU side",
            "seconds_per_step": total, "phase_seconds": phases}


def self_launch(a):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a CHILD `python -m torch.distributed.run` (a fresh
    process tree: this parent has not initialised the GPU and never does), stream its stderr through, forward the ranks' one
    JSON line on stdout and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("[bench] the ranks exited without printing a result line", file=sys.stderr)
        return 1
    return proc.returncode


def launch_check(a, world, rank, backend):
    """The distributed skeleton of the bench without the engine: barrier, K timed no-op steps, max over ranks, one JSON line."""
    import torch.distributed as dist
    dev = "cpu" if backend == "gloo" else f"cuda:{torch.cuda.current_device()}"
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(got, tt)
        per_rank = [float(g.item()) for g in got]
    if rank == 0:
        print(json.dumps({"metric": "GRPO rollout samples/sec", "value": 0.0, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": max(per_rank) / max(a.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "launch_check": True, "dist_backend": backend if world > 1 else None,
                          "rccl_ranks": world if backend == "nccl" and world > 1 else 0,
                          "rank_ms_per_step": [t / max(a.steps, 1) * 1e3 for t in per_rank]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    t_start = time.perf_counter()
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: reporting n_gpus={world}", file=sys.stderr)
    backend = os.environ.get("VR2_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; gloo only for single-GPU / CPU rehearsals
    if a.launch_check and (backend == "gloo" or world == 1):
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        return launch_check(a, world, rank, "gloo")
    local = local % max(1, torch.cuda.device_count())       # (rehearsals put several ranks on one GPU; a node has one GPU per rank)
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    if a.launch_check:
        return launch_check(a, world, rank, backend)
    from video_r2_amd import lib, weights as W
    from video_r2_amd.engine import Engine
    from video_r2_amd.host import grpo as hg
    lib.load()
    cfg = model_cfg(a.model)
    G, C = a.G, a.max_new
    # random-init weights of the named architecture (no checkpoints offline); the frozen reference model is a second,
    # separately stored copy of the LM (as the reference trainer keeps one: grpo_trainer.py:329-346)
    sd = W.random_state_dict(cfg, seed=0, device=device)
    vis = W.build_vision(cfg, sd, device)
    pol = W.build_text(cfg, sd, device)
    ref = W.build_text(cfg, sd, device)
    del sd
    torch.cuda.empty_cache()
    eng = Engine(cfg, vis, pol, device=device)
    fp8 = a.decode_weights in ("fp8", "fp8w")
    if fp8:
        W.add_fp8_decode_copies(pol)
        eng.decode_fp8 = True
        eng.decode_fp8_act = a.decode_weights == "fp8"
        if a.decode_weights == "fp8" and not a.no_gemm_fp8:
            W.add_fp8_decode_copies(ref)                           # frozen: quantised once
            eng.gemm_fp8 = True
    from video_r2_amd.train import Trainer
    upd = None if a.no_update else Trainer(eng, pol, grad_exchange=a.grad_exchange)       # fp32 masters + Adam moments + transposed packed copies
    P = a.prompts_per_gpu
    if P is None:          # 176 sequences per decode weight pass up to 32 frames (22 groups of G = 8, 44 of G = 4); 64-frame prompts and
        P = max(1, 176 // G) if (a.decode_weights != "fp8w" and a.frames <= 32) else max(1, 128 // G)     # weight-only fp8 keep 128
    P = max(1, min(P, (128 if a.decode_weights == "fp8w" else 256) // G))
    # synthetic prompts resident before the timed region: distinct ones for every step while they fit --prompt-pool-gb (19 MB of fp32
    # pixel rows per 32-frame prompt), else a pool of that many steps' worth walked cyclically (every step still runs all of its
    # work: nothing is cached between steps, and the weights have moved by the time a prompt comes round again)
    need_steps = a.steps + a.warmup + 1                     # + one instrumented step for the phase split
    per_step_bytes = P * (a.frames // 2) * 256 * 1176 * 4
    pool_steps = max(2, min(need_steps, int(a.prompt_pool_gb * 1e9 // max(1, per_step_bytes))))
    n_prompts = pool_steps * P + 2                          # + 2 recipe-shaped steps
    prompts = [synth_prompt(cfg, a.frames, rank * 1000 + i, device) for i in range(n_prompts)]
    slots = eng.prefix_slots(P, max(len(p[2]) for p in prompts), tag="pol")
    torch.cuda.synchronize()

    def step(i, rec=None, mine=None, lens=None):
        """One GRPO step over this rank's P prompt groups.  rec(name) marks a phase boundary (instrumented extra step only).
        lens (host ints, one per sequence): emulated EOS positions -- see --eos-mix (Engine.decode finish_at=)."""
        mark = rec if rec is not None else (lambda name: None)
        mine = prompts[(i % pool_steps) * P:(i % pool_steps + 1) * P] if mine is None else mine
        Pn = len(mine)

        # the step's P clips go through the vision tower in ONE pass and its P prompts through ONE policy prefill and ONE
        # reference prefill (M = P x 4096 patches / P x S tokens: every projection runs on full 256-row tiles)
        mark(None)
        enc = eng.encode_videos([dict(pixel_values=px, grid=grid) for px, grid, _ in mine])
        mark("vit")
        states = eng.prefill_batch(pol, [(ids, enc[p_][0], enc[p_][1], [1.0]) for p_, (_, _, ids) in enumerate(mine)], slots[:Pn], tag="pol")
        mark("prefill")
        toks_all, _ = eng.decode(pol, states, G, C, temperature=0.9, top_k=50, seed=1000 * rank + i, eos_token_id=None,
                                 use_graph=not a.no_graph, finish_at=lens)      # lens: the decode stops at its first 64-step check
        Cd = int(toks_all.shape[1])                                             # past the longest sequence's (emulated) EOS
        mark("decode")
        eng.sync()
        comp_all = toks_all.cpu().numpy()                              # the rewards are host functions of the completions: ONE copy
        rew_all = torch.from_numpy(stub_rewards(comp_all)).to(device)  # per step, no host sync inside the loop over the groups
        losses = []
        for p_, (px, grid, ids) in enumerate(mine):
            emb, plan = enc[p_]
            toks = toks_all[p_ * G:(p_ + 1) * G]
            mark(None)
            ln = None if lens is None else np.minimum(lens[p_ * G:(p_ + 1) * G], Cd)
            # reference log-probs over the unified [prompt | completions] row space: the row space and launches of the policy's
            # training forward below, so identical weights give identical bits (Engine.score_unified)
            lp_ref = eng.score_unified(ref, ids, emb, plan, [1.0], toks, 0.9, tag="ref", lens=ln).clone()   # (a workspace: the next pass reuses it)
            mark("ref_prefill+ref_score")
            lp = (eng.score_unified(pol, ids, emb, plan, [1.0], toks, 0.9, tag="ref", lens=ln, allow_fp8=False) if upd is None
                  else upd.forward(ids, emb, plan, [1.0], toks, 0.9, lens=ln))
            mark("policy_forward")
            with torch.cuda.stream(eng.stream):
                mask, _, _ = hg.completion_mask(toks, -1)
                if ln is not None:
                    mask = (torch.arange(Cd, device=device)[None, :] < torch.from_numpy(ln).to(device)[:, None]).int()
                adv = hg.group_advantages(rew_all[p_ * G:(p_ + 1) * G], [1.0], G, loss_type="bnpo")["advantages"]
                out = hg.grpo_loss(lp, mask, adv, ref_logps=lp_ref, beta=0.04, loss_type="bnpo", stream=eng.s)
            if upd is not None:
                upd.backward(out["dlogp"], exchange=(p_ == Pn - 1))    # gradients accumulate over the groups; the last backward
                                                                       # puts each layer's bucket on the wire as it completes
            mark("loss+backward")
            losses.append(out["loss"])
        mark(None)
        if upd is not None:
            upd.optimizer_step(lr=1e-6, weight_decay=0.01, max_grad_norm=1.0, grad_scale=1.0 / Pn)   # + RCCL all-reduce for N>1
            if fp8:
                with torch.cuda.stream(eng.stream):
                    W.add_fp8_decode_copies(pol)                     # the next rollout decodes the updated policy
        mark("clip+adamw(+allreduce)")
        eng.sync()
        return float(sum(float(l.item()) for l in losses)) / Pn

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def say(msg):        # progress on stderr (rank 0): a run of several minutes must not look hung to whoever launched it
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    say(f"weights, optimiser state and {n_prompts} synthetic prompts resident; {a.warmup} warm-up + {a.steps} timed steps of {P} groups")
    for i in range(a.warmup):
        step(i)
        say(f"warm-up step {i + 1}/{a.warmup} done")
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
        if (i + 1) % 4 == 0 or i + 1 == a.steps:
            say(f"timed step {i + 1}/{a.steps} done ({(time.perf_counter() - t0) / (i + 1):.2f} s per step so far)")
    barrier()
    dt = time.perf_counter() - t0
    rank_dt = [dt]
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        got = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(got, tt)
        rank_dt = [float(g.item()) for g in got]
        dt = max(rank_dt)                                   # the job is as slow as its slowest rank

    # ---- the reference recipe's per-device shape: ONE prompt group per GPU per optimiser step (train_grpo.sh:4-7:
    # per_device_train_batch_size 1 -> G rows per decode weight pass).  Untimed for `value`; reported beside it.
    recipe = None
    say("timed region done; untimed extras follow (recipe shape, emulated EOS mix, instrumented step, roofline kernels)")
    if not a.no_recipe_shape and P > 1:
        step(0, mine=prompts[-2:-1])                           # captures the G-row decode graph
        barrier()
        t1 = time.perf_counter()
        step(0, mine=prompts[-1:])
        barrier()
        recipe = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            tr = torch.tensor([recipe], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            recipe = float(tr.item())

    # ---- EOS-terminated completions, emulated (--eos-mix): the same step with per-sequence lengths; the decode stops at the
    # longest sequence and the scoring / training passes run on packed rows.  Untimed for `value`; reported beside it.
    eos_mix = None
    if a.eos_mix and a.eos_mix != "off":
        kind, lo, hi = a.eos_mix.split(":")
        assert kind == "uniform", "--eos-mix uniform:LO:HI"
        rng = np.random.default_rng(99 + rank)
        hi = min(int(hi), C)
        lo = max(1, min(int(lo), hi))                          # (short token budgets of rehearsal runs: clamp into [1, C])
        lens_a = rng.integers(lo, hi + 1, size=P * G)
        lens_b = rng.integers(lo, hi + 1, size=P * G)
        step(0, lens=lens_a)                                   # captures the decode graph with the finished-sequence skip
        say("eos-mix warm-up step done")
        barrier()
        t1 = time.perf_counter()
        step(1, lens=lens_b)
        barrier()
        em = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            te = torch.tensor([em], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            em = float(te.item())
        eos_mix = {"lengths": f"uniform {lo}..{hi} tokens per sequence (seeded), mean {float(lens_b.mean()):.0f}, longest {int(lens_b.max())}",
                   "samples_per_s": G * P * world / em, "ms_per_step": em * 1e3,
                   "note": "full step at the default workload with emulated EOS positions: decode until the longest sequence ends "
                           "(finished sequences stop walking their own keys), scoring and training on the packed rows (sum of "
                           "lengths instead of G x C); one sample"}

    # ---- phase split + roofline of the dominant kernel, HIP events on the engine's stream (rank 0)
    say("extras done; instrumented step")
    if rank != 0:
        step(a.warmup + a.steps)      # the instrumented step below runs on EVERY rank: its optimiser step is a collective
    extra = {}
    roof = None
    if rank == 0:
        import ctypes
        L = lib.load()

        def ev():
            e = ctypes.c_void_p()
            L.vr2_event_create(ctypes.byref(e))
            return e

        def elapsed(e0, e1):
            ms = ctypes.c_float()
            L.vr2_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            return ms.value

        # one more (untimed) step with HIP events at the phase boundaries, on the engine's stream; summed over the P groups
        evs, last = [], [None]

        def rec(name):
            e = ev()
            L.vr2_event_record(e, ctypes.c_void_p(eng.s))
            if name is not None and last[0] is not None:
                evs.append((name, last[0], e))
            last[0] = e

        step(a.warmup + a.steps, rec)
        eng.sync()
        ph = {}
        for name, e0_, e1_ in evs:
            ph[name] = ph.get(name, 0.0) + elapsed(e0_, e1_)
        t = cfg.text
        H, I, V, Lh = t.hidden_size, t.intermediate_size, t.vocab_size, t.num_hidden_layers
        nqkv = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
        w_bytes = 2 * (Lh * (nqkv * H + H * H + 3 * I * H) + V * H)
        kv_tok = 2 * 2 * t.num_key_value_heads * t.head_dim * Lh
        S = len(prompts[-1][2])
        NG = P * G
        kv_bytes = kv_tok * (P * S + NG * (C / 2))    # per token: each group's prefix once + half of every sequence's suffix on average
        step_ms = ph["decode"] / max(C - 1, 1)
        extra = {"phase_ms": ph,
                 "rollout_samples_per_s": NG / ((ph["vit"] + ph["prefill"] + ph["decode"]) * 1e-3),
                 "decode_rows_per_weight_pass": NG,
                 "decode_ms_per_token": step_ms,
                 "decode_step_GBps": (w_bytes + kv_bytes) / (step_ms * 1e-3) / 1e9,
                 "vit_prefill_TFLOPs": P * (1.285e9 * 128 * a.frames + (2 * 6.526e9 / 28 * Lh) * S) / ((ph["vit"] + ph["prefill"]) * 1e-3) / 1e12 if a.model == "7b" else None}
        # dominant kernel of the decode phase: the gate/up projection + SiLU*mul GEMV at the step's row count (NG rows per
        # weight pass): 2*I*H bf16 weight bytes per launch
        xn = eng.buf("d.xn", (NG, H), torch.bfloat16)
        act = eng.buf("d.act", (NG, I), torch.bfloat16)
        reps = 3
        e0, e1 = ev(), ev()
        fp8r = fp8 and NG <= (256 if eng.decode_fp8_act else eng.decode_fp8_max_rows)   # above that the engine decodes with the bf16 kernels
        a8 = fp8r and eng._a8w8(pol, "l0.gu_w", NG)           # fp8 activations on the fp8 MFMA (> 16 rows)
        if a8:
            xq8, xqs = eng._quant(xn, NG, H)                  # quantised once, outside the timed launches

        def gu(l):
            if a8:
                q8, sc = pol.f8[f"l{l}.gu_w"]
                lib.call("vr2_gemv_gateup_silu_a8w8", xq8, xq8.stride(0), xqs, NG, q8, sc, I, H, act, act.stride(0), eng.s)
            elif fp8r:
                q8, sc = pol.f8[f"l{l}.gu_w"]
                lib.call("vr2_gemv_gateup_silu_w8", xn, xn.stride(0), NG, q8, sc, I, H, act, act.stride(0), eng.s)
            else:
                lib.call("vr2_gemv_gateup_silu_bf16", xn, xn.stride(0), NG, pol[f"l{l}.gu_w"], I, H, act, act.stride(0), eng.s)
        def timed_launches(launch):
            """average duration of launch(l), l over the Lh layers' distinct weight buffers, HIP events on the engine's stream.
            The launches are captured into ONE hipGraph and the graph is replayed -- the way the decode step issues them
            (Engine.decode) -- so the figure is the kernel's duration plus the in-graph dependency gap, not the 3-4 us an eager
            launch loop adds per kernel; `--no-graph`, or a capture failure, falls back to the eager loop.  -> (ms, how)"""
            for l in range(Lh):   # untimed pass (first touch, code objects and launch attributes loaded before any capture)
                launch(l)
            eng.sync()
            gex = None
            if not a.no_graph:
                gex = ctypes.c_void_p()
                rc = L.vr2_graph_begin(ctypes.c_void_p(eng.s))
                if rc == 0:
                    try:
                        for l in range(Lh):
                            launch(l)
                    finally:
                        rc = L.vr2_graph_end(ctypes.c_void_p(eng.s), ctypes.byref(gex))
                if rc != 0 or not gex.value:
                    gex = None
            if gex is not None:
                L.vr2_graph_launch(gex, ctypes.c_void_p(eng.s))          # one untimed replay
            L.vr2_event_record(e0, ctypes.c_void_p(eng.s))
            for _ in range(reps):
                if gex is not None:
                    L.vr2_graph_launch(gex, ctypes.c_void_p(eng.s))
                else:
                    for l in range(Lh):
                        launch(l)
            L.vr2_event_record(e1, ctypes.c_void_p(eng.s))
            eng.sync()
            ms = elapsed(e0, e1) / (reps * Lh)
            if gex is not None:
                L.vr2_graph_destroy(gex)
            return ms, ("hipGraph replay of %d launches" % Lh if gex is not None else "eager launch loop")

        k_ms, k_how = timed_launches(gu)
        alg = 2 * I * H * (1 if fp8r else 2) + (2 * I * 4 if fp8r else 0) + NG * H * (1 if a8 else 2) + NG * I * 2
        ach = alg / (k_ms * 1e-3) / 1e9
        mbs = 2 if NG <= 32 else 4 if NG <= 64 else 8 if NG <= 128 else 10 if NG <= 160 else 12 if NG <= 192 else 14 if NG <= 224 else 16
        if a8:
            kname = "gemv_ring8_kernel<1, %d, 1> (fp8 activations x fp8 weights, v_mfma_f32_16x16x32_fp8_fp8)" % mbs
        elif fp8r:
            kname = "gemv16_w8_kernel<1>" if NG <= 16 else ("gemv_lds_w8_kernel<1, %d>" % mbs if NG <= 64 else "gemv_lds_w8s_kernel<1, 8, 1>")
        else:
            kname = "gemv16_kernel<1, 1, 4>" if NG <= 16 else "gemv_ring_kernel<1, %d, 1, 0>" % mbs
        # HBM bytes per launch from rocprofv3 PMC (profiles/pmc_traffic.json; null once the kernel's sources have changed)
        pm = None
        if a.model == "7b":
            pm = pmc_entry("gemv_gateup_bf16_%drows_7b" % NG) if not fp8r else (pmc_entry("gemv_gateup_w8_128rows_7b") if (not a8 and NG == 128) else None)
        extra["roofline_decode"] = {"bound": "hbm", "kernel": kname + " (decode gate/up GEMV + SiLU*mul, %d rows)" % NG, "achieved": ach,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                    "traffic": pm["traffic_bytes"] if pm else None, "traffic_source": pm["source"] if pm else None,
                                    "avg_launch_us": k_ms * 1e3, "algorithmic_bytes_per_launch": alg, "timed_as": k_how}
        # the split-K instantiation of the same kernel family: the down projection (K = I, fp32 partial slabs summed by the
        # next kernel), at the split count the engine's decode step uses; and the FAMILY: q/k/v + o + down of every layer plus
        # the lm_head, as one decode step issues them -- weights + activations in + slabs out over their summed time
        if not fp8r and NG > 16:
            db = eng._decode_bufs(NG, C, S, eng._prefix_chunk(P, S))
            p_d, ns_d = db["p_d"], db["p_d"].shape[0]
            p_q, ns_q, p_o, ns_o = db["p_qkv"], db["p_qkv"].shape[0], db["p_o"], db["p_o"].shape[0]
            att_b, lg_b = db["att"], db["logits"]

            def dn(l):
                lib.call("vr2_gemv_bf16", act, act.stride(0), NG, pol[f"l{l}.down_w"], H, pol.shape[f"l{l}.down_w"][1], p_d,
                         p_d.stride(0) if ns_d > 1 else 0, p_d.stride(-2), ns_d, eng.s)
            d_ms, _ = timed_launches(dn)
            d_alg = 2 * I * H + NG * I * 2 + ns_d * NG * H * 4
            pd = pmc_entry("gemv_down_bf16_%drows_7b" % NG) if a.model == "7b" else None
            extra["roofline_decode_splitk"] = {"bound": "hbm", "kernel": "gemv_ring_kernel<0, %d, 1, 0> (decode down projection, %d rows, %d k-parts)" % (mbs, NG, ns_d),
                                               "achieved": d_alg / (d_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": d_alg / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pd["traffic_bytes"] if pd else None,
                                               "traffic_source": pd["source"] if pd else None,
                                               "avg_launch_us": d_ms * 1e3, "algorithmic_bytes_per_launch": d_alg}

            def fam(l):
                eng._gemv(xn, pol, f"l{l}.qkv_w", NG, p_q, ns_q)
                eng._gemv(att_b, pol, f"l{l}.o_w", NG, p_o, ns_o)
                eng._gemv(act, pol, f"l{l}.down_w", NG, p_d, ns_d)
                if l == Lh - 1:
                    eng._gemv(xn, pol, "lm_head", NG, lg_b, 1)
            f_ms, _ = timed_launches(fam)                      # per layer: 3 launches (+ the lm_head once per Lh layers)
            qd = t.num_attention_heads * t.head_dim
            f_alg = (Lh * (2 * (nqkv * H + H * qd + H * I) + NG * 2 * (H + qd + I) + NG * 4 * (ns_q * nqkv + ns_o * H + ns_d * H))
                     + 2 * V * H + NG * 2 * H + NG * 4 * V)
            f_tot = f_ms * Lh
            extra["roofline_decode_splitk_family"] = {
                "bound": "hbm", "kernel": "gemv_ring_kernel<0, %d, 1, 0> family of one decode step: %d x (q/k/v %d parts, o %d parts, down %d parts) + lm_head, %d rows"
                                          % (mbs, Lh, ns_q, ns_o, ns_d, NG),
                "achieved": f_alg / (f_tot * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f_alg / (f_tot * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None, "step_total_us": f_tot * 1e3, "launches": 3 * Lh + 1, "algorithmic_bytes_per_step": f_alg}

        # the largest MFMA-bound kernel of the training phases, measured the same way: the gate/up GEMM + SwiGLU epilogue over
        # the unified row space of one prompt group (T = S + G*C rows), one launch per layer
        Tr = S + G * C
        xa = torch.randn(Tr, H, device=device).to(torch.bfloat16)
        ya = torch.empty(Tr, I, dtype=torch.bfloat16, device=device)
        n_gu = pol.shape["l0.gu_w"][0]
        for l in range(Lh):
            lib.call("vr2_gemm_bf16", xa, H, pol[f"l{l}.gu_w"], None, ya, I, Tr, n_gu, H, 4, eng.s)
        L.vr2_event_record(e0, ctypes.c_void_p(eng.s))
        for l in range(Lh):
            lib.call("vr2_gemm_bf16", xa, H, pol[f"l{l}.gu_w"], None, ya, I, Tr, n_gu, H, 4, eng.s)
        L.vr2_event_record(e1, ctypes.c_void_p(eng.s))
        eng.sync()
        g_ms = elapsed(e0, e1) / Lh
        g_tf = 2.0 * Tr * n_gu * H / (g_ms * 1e-3) / 1e12
        tm_, tn_ = -(-Tr // 256), -(-n_gu // 256)
        use256 = tm_ * tn_ * Tr * 4 >= 3 * -(-tm_ * tn_ // 256) * 256 * tm_ * 256          # the launcher's rule (csrc/gemm.hip)
        gname = "gemm256_kernel<4>" if use256 else "gemm_kernel<4, %d>" % (4 if -(-Tr // 128) * -(-n_gu // 128) >= 512 else 2)
        # THE dominant kernel of the step (largest share of kernel time in the tracked rocprofv3 summary): `roofline`
        pg = pmc_entry("gemm256_gateup_swiglu_%drows_7b" % Tr) if (a.model == "7b" and use256) else None
        roof = {"bound": "mfma", "kernel": gname + " (gate/up GEMM + SwiGLU of the scoring / training passes, %d x %d x %d)" % (Tr, n_gu, H),
                "achieved": g_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g_tf / MFMA_PEAK_TFLOPS,
                "traffic": pg["traffic_bytes"] if pg else None, "traffic_source": pg["source"] if pg else None,
                "mfma_busy_pct": pg.get("mfma_busy_pct") if pg else None,
                "algorithmic_bytes_per_launch": 2.0 * (Tr * H + n_gu * H + Tr * (n_gu // 2)),
                "avg_launch_us": g_ms * 1e3, "algorithmic_flops_per_launch": 2.0 * Tr * n_gu * H,
                "dominant_by": "profiles/r04_kernel_stats_default_bench_p22.md (largest share of the step's kernel time)"}
        # the weight-gradient GEMM of the same layer (dW[2I][H] += dY[T][2I]^T X[T][H]): the largest launch of the backward
        if hasattr(upd, "state") and "l0.gu_w" in getattr(upd.state, "grad", {}):
            # operands = the step's own buffers (dY: the gate/up pre-activation gradients the last backward left in its workspace,
            # X: the saved normed inputs of layer 0): allocated at start-up like every launch of the step.  Operands allocated
            # HERE, with 250 GB already in use, ran the same launch 20 % slower (2.72 vs 2.25 ms) -- late allocations come from
            # fragmented physical memory -- while the step's own launches average the isolated time under rocprofv3
            I16 = (I + 15) // 16 * 16
            dya = eng.buf("t.dgu", (Tr, 2 * I16), torch.bfloat16) if 2 * I16 == n_gu else torch.randn(Tr, n_gu, device=device).to(torch.bfloat16)
            xw = eng.buf("t.xn2", (Lh, Tr, H), torch.bfloat16)[0]
            gw_ = upd.state.grad["l0.gu_w"]
            keep = gw_.clone()
            wreps = 3
            lib.call("vr2_gemm_tn_bf16", dya, dya.stride(0), xw, xw.stride(0), gw_, gw_.stride(0), n_gu, H, Tr, 1, eng.s)
            L.vr2_event_record(e0, ctypes.c_void_p(eng.s))
            for _ in range(wreps):
                lib.call("vr2_gemm_tn_bf16", dya, dya.stride(0), xw, xw.stride(0), gw_, gw_.stride(0), n_gu, H, Tr, 1, eng.s)
            L.vr2_event_record(e1, ctypes.c_void_p(eng.s))
            eng.sync()
            w_ms = elapsed(e0, e1) / wreps
            gw_.copy_(keep)
            w_tf = 2.0 * Tr * n_gu * H / (w_ms * 1e-3) / 1e12
            extra["roofline_wgrad"] = {"bound": "mfma", "kernel": "gemm_tn256_kernel<1> (+ <2>: last round as k-parts) (gate/up weight gradient, %d x %d x %d, fp32 +=)" % (n_gu, H, Tr),
                                       "achieved": w_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": w_tf / MFMA_PEAK_TFLOPS, "traffic": None,
                                       "avg_launch_us": w_ms * 1e3, "algorithmic_flops_per_launch": 2.0 * Tr * n_gu * H}
            del keep
        del xa, ya

    if rank == 0:
        samples = G * P * a.steps * world
        line = {
            "metric": "GRPO rollout samples/sec", "value": samples / dt, "unit": "samples/s", "n_gpus": world,
            "value_is": "completions / wall second of the FULL step (rollout + reference and policy scoring + loss + backward + "
                        f"all-reduce + AdamW) at {P} prompt groups per GPU per optimiser step; rollout-only rate: rollout_samples_per_s",
            "rccl_ranks": world if (world > 1 and backend == "nccl") else 0, "dist_backend": backend if world > 1 else None,
            "rank_ms_per_step": [t / a.steps * 1e3 for t in rank_dt],
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not fp8 else (("bf16 training; decode, rollout prefill and reference scoring: fp8 e4m3 weights per channel x fp8 activations per row on the fp8 MFMA" if eng.gemm_fp8
                                                                             else "bf16 (decode: fp8 e4m3 weights per channel x fp8 activations per row on the fp8 MFMA)") if a.decode_weights == "fp8"
                                                                            else "bf16 (decode weights fp8 e4m3, per-channel scales)"), "data": "synthetic",
            "config": {"workload": f"Qwen2.5-VL-{a.model.upper()} {a.frames} frames 224^2, G={G}, {C} new tokens, {P} prompt group(s) per GPU per step",
                       "prompts_per_gpu_per_step": P, "prompt_pool_steps": pool_steps,
                       "prompt_tokens": 32 * a.frames + 64, "weights": "random-init seed 0, bf16",
                       "phases_timed": ["vit", "prefill", "decode(G x C, sampling T=0.9 top-k 50, EOS off)",
                                        "ref prefill + ref log-probs", "policy training forward", "advantages + GRPO loss"]
                                       + ([] if a.no_update else ["backward", "grad all-reduce (N>1)", "clip + AdamW"]),
                       "parallelism": f"dp{world} (prompt groups sharded; one gradient all-reduce per step, {a.grad_exchange} on the wire)"},
            "last_loss": loss,
            "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),   # torch allocator high-water mark on rank 0
            "hbm_reserved_GB": round(torch.cuda.max_memory_reserved() / 1e9, 1),  # ... incl. the allocator's cached blocks
            "hbm_free_GB": round(torch.cuda.mem_get_info()[0] / 1e9, 1),          # what the driver still reports free at the end
        }
        if recipe is not None:
            line["recipe_shape_samples_per_s"] = G * world / recipe        # 1 group per GPU per step, full step, one sample of it
            line["recipe_shape_ms_per_step"] = recipe * 1e3
        if eos_mix is not None:
            line["eos_mix"] = eos_mix
        line.update(extra)
        if roof:
            line["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            if a.cpu_baseline == "full":
                say("CPU baseline: the oracle runs the whole step on the host cores")
                line["cpu_baseline"] = cpu_baseline_full(a.model, a.frames, G, C, P)
            else:
                say("CPU baseline (bounded sample of the oracle)")
                line["cpu_baseline"] = cpu_baseline(a.model, a.frames, G, C)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                 # rank 0 times its roofline kernels after the last collective: the others wait here, so the
        dist.destroy_process_group()   # communicator is torn down by all ranks together


if __name__ == "__main__":
    main()
