import sys, ctypes, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
exec(open('profiles/src/tune_gemv_m.py').read().split("NL = 28")[0])
N, K = 37888, 3584
Ws = [torch.empty(((N + 15) // 16, K // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
for M in (64, 128):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
    def fn():
        for W in Ws:
            lib.call("vr2_gemv_gateup_silu_bf16", x, K, M, W, N // 2, K, act, N // 2, s)
    us = timed_graph(fn) * 1e3 / 28
    print(f"gateup M={M}: {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s", flush=True)
