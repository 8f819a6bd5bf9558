"""Training-shape attention kernels in isolation: 7B heads (28 q / 4 kv, hd 128), prompt S, G completions of C rows.
Prints the time per launch of the forward (vr2_attn_tok with LSE), dQ and dK/dV kernels and the wgrad GEMM shapes."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
from video_r2_amd.host import vision_plan as vp
lib.load()
dev = "cuda:0"
S_, G, C = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1088, 8, 1024)))
nH, nKV, hd = 28, 4, 128
group = nH // nKV
T = S_ + G * C
scale = hd ** -0.5
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(T, nH * hd, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(T, nKV * hd, device=dev, generator=g).to(torch.bfloat16)
v = torch.randn(T, nKV * hd, device=dev, generator=g).to(torch.bfloat16)
do = torch.randn(T, nH * hd, device=dev, generator=g).to(torch.bfloat16)
o = torch.zeros_like(q); lse = torch.zeros(T, nH, device=dev)
work = np.concatenate([vp.causal_work(S_, 0, 0, 0, 0)] + [vp.causal_work(C, S_ + i * C, 0, S_, S_ + i * C) for i in range(G)])
dwork = torch.from_numpy(work).to(dev)
wq = np.concatenate([vp.causal_work(S_, 0, 0, 0, 0, q_tile=128)] + [vp.causal_work(C, S_ + i * C, 0, S_, S_ + i * C, q_tile=128) for i in range(G)])
dwq = torch.from_numpy(np.ascontiguousarray(vp.heaviest_first(wq) if len(sys.argv) <= 4 else wq)).to(dev)
rowc = torch.zeros(2, nH, T, device=dev)
dq = torch.zeros_like(q); dk = torch.zeros(T, nKV * hd, device=dev); dv = torch.zeros_like(dk)
pairs = S_ * (S_ + 1) / 2 + G * (C * S_ + C * (C + 1) / 2)
fl = 4 * pairs * nH * hd

def timeit(name, fn, flops, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} {ms*1e3:9.1f} us   {flops/ms/1e9:8.1f} TFLOP/s", flush=True)

timeit("attn_tok fwd (+lse)", lambda: lib.call("vr2_attn_tok", dwq, dwq.shape[0], q, nH * hd, k, v, k, v, nKV * hd, o, nH * hd, nH, group, hd, scale, lse, st), fl)
timeit("attn_delta", lambda: lib.call("vr2_attn_delta", o, do, nH * hd, T, nH, hd, lse, rowc, st), 0)
timeit("attn_bwd_dq", lambda: lib.call("vr2_attn_bwd_dq", dwq, dwq.shape[0], q, nH * hd, k, v, k, v, nKV * hd, do, rowc, T, dq, nH, group, hd, scale, st), fl * 1.5)
for qs, ms_ in ((512, 64), (1024, 64), (512, 32), (1024, 128)):
    kw = torch.from_numpy(vp.key_work(S_, G, C, q_split=qs, group=group, max_steps=ms_)).to(dev)
    timeit(f"attn_bwd_dkv q_split={qs} max_steps={ms_} items={kw.shape[0]}",
           lambda: lib.call("vr2_attn_bwd_dkv", kw, kw.shape[0], q, nH * hd, k, v, nKV * hd, do, rowc, T, dk, dv, nH, group, hd, scale, st), fl * 2.5)
# wgrad shapes of one layer at T rows: dW[out][in] += dY^T X
for (M, N) in ((4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944)):
    a = torch.randn(T, M, device=dev, generator=g).to(torch.bfloat16)
    b_ = torch.randn(T, N, device=dev, generator=g).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    timeit(f"wgrad {M}x{N} (+=)", lambda: lib.call("vr2_gemm_tn_bf16", a, M, b_, N, c, N, M, N, T, 1, st), 2.0 * T * M * N, n=5)
    del a, b_, c
