"""Decode attention in isolation at the bench's batch: P = 16 prompt groups x 8 sequences, 28 q / 4 kv heads, S = 1088 prompt keys,
C = 1024 suffix slots, `step` live suffix keys; one launch per layer over 28 distinct KV buffers (nothing cache-resident).
Usage: python profiles/src/attn_decode_bench.py [step ...]
  VR2_PREFIX_CHUNK / VR2_DECODE_CHUNK: chunk lengths, ';'-separated lists are swept ("320;192", "512;256")
  VR2_DEPTHS: ';'-separated "suffix,prefix" tiles-in-flight settings (vr2_tune_set keys 3 / 4), e.g. "1,1;2,1;3,2"
Every setting's output is compared with the first setting's at the same step and chunks (max abs difference printed)."""
import os, sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
L_ = lib.load()
dev = "cuda:0"
st_ = torch.cuda.current_stream().cuda_stream
P, spg, nH, nKV, hd, S_, C, L = 16, 8, 28, 4, 128, 1088, 1024, 28
G, group = P * spg, nH // nKV
pchunks = [int(v) for v in os.environ.get("VR2_PREFIX_CHUNK", "320").split(";")]
schunks = [int(v) for v in os.environ.get("VR2_DECODE_CHUNK", "512").split(";")]
depths = [tuple(int(x) for x in v.split(",")) for v in os.environ.get("VR2_DEPTHS", "1,1").split(";")]
g = torch.Generator(device=dev).manual_seed(0)
kp = [torch.randn(P * S_, nKV * hd, device=dev, generator=g).to(torch.bfloat16) for _ in range(L)]
vp = [torch.randn(P * S_, nKV * hd, device=dev, generator=g).to(torch.bfloat16) for _ in range(L)]
ks = [torch.randn(G * C, nKV * hd, device=dev, generator=g).to(torch.bfloat16) for _ in range(L)]
vs = [torch.randn(G * C, nKV * hd, device=dev, generator=g).to(torch.bfloat16) for _ in range(L)]
q = torch.randn(G, nH * hd, device=dev, generator=g).to(torch.bfloat16)
plen = torch.full((P,), int(os.environ.get('VR2_PLEN', S_)), dtype=torch.int32, device=dev)   # live prompt keys (<= S_)
steps = [int(a) for a in sys.argv[1:]] or [15, 255, 511, 767, 1023]
for pchunk in pchunks:
    for schunk in schunks:
        npc, nsc = -(-S_ // pchunk), -(-C // schunk)
        po = torch.zeros(G * nH * (npc + nsc), hd, device=dev); pml = torch.zeros(G * nH * (npc + nsc), 2, device=dev)
        out = torch.zeros(G, nH * hd, dtype=torch.bfloat16, device=dev)
        ref = {}
        avg = {}
        for ds, dp in depths:
            L_.vr2_tune_set(3, ds); L_.vr2_tune_set(4, dp)
            for step in steps:
                stp = torch.tensor([step], dtype=torch.int32, device=dev)

                def run():
                    for l in range(L):
                        lib.call("vr2_attn_decode", q, nH * hd, kp[l], vp[l], ks[l], vs[l], nKV * hd, plen, spg, S_ * nKV * hd, C, stp, schunk, pchunk,
                                 npc, nsc, G, nH, group, hd, hd ** -0.5, po, pml, out, nH * hd, None, st_)
                run(); torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): run()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (3 * L)
                mb = (P * S_ + G * (step + 1)) * nKV * hd * 2 * 2 / 1e6
                o = out.float().clone()
                diff = (o - ref.setdefault(step, o)).abs().max().item()
                avg.setdefault((ds, dp), []).append(us)
                print(f"pchunk {pchunk} schunk {schunk} depth s{ds} p{dp} step {step:5d}: {us:7.2f} us per launch (attention + combine), "
                      f"{mb:7.1f} MB of K/V = {mb / us:6.2f} TB/s, max |out - first setting| {diff:.1e}", flush=True)
        for k, v in avg.items():
            print(f"== pchunk {pchunk} schunk {schunk} depth s{k[0]} p{k[1]}: mean over steps {sum(v) / len(v):.2f} us", flush=True)
