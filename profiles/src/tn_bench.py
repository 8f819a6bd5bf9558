import sys, ctypes, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
exec(open('profiles/src/tune_gemv_m.py').read().split("NL = 28")[0])
T = 9280
for (M, N, name) in [(37888, 3584, "gate/up"), (3584, 18944, "down"), (4608, 3584, "qkv"), (3584, 3584, "o")]:
    A = torch.randn(T, M, device=dev).to(torch.bfloat16); B = torch.randn(T, N, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    for acc in (0, 1):
        def fn():
            lib.call("vr2_gemm_tn_bf16", A, M, B, N, C, N, M, N, T, acc, s)
        ms = timed_graph(fn, reps=5)
        print(f"wgrad {name:8s} M={M} N={N} K={T} acc={acc}: {ms*1e3:8.1f} us  {2.0*M*N*T/ms/1e9:7.1f} TF/s", flush=True)
