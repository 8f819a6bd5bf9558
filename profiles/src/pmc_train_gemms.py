"""The training-step GEMMs at T = 9280 rows (NT gate/up + SwiGLU, NT q/k/v, TN wgrads) for PMC MFMA-utilisation collection."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
lib.load()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream().cuda_stream
T, K = 9280, 3584
x = torch.randn(T, K, device=dev).to(torch.bfloat16)
for N, epi in ((37888, 4), (4608, 1)):
    Ws = [torch.empty(((N + 15) // 16, K // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(3)]
    out = torch.empty(T, N // 2 if epi == 4 else N, dtype=torch.bfloat16 if epi == 4 else torch.float32, device=dev)
    for rep in range(2):
        for W in Ws:
            lib.call("vr2_gemm_bf16", x, K, W, None, out, out.stride(0), T, N, K, epi, s)
for (M, N) in ((37888, 3584), (3584, 18944), (4608, 3584)):
    A = torch.randn(T, M, device=dev).to(torch.bfloat16); B = torch.randn(T, N, device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    for rep in range(3):
        lib.call("vr2_gemm_tn_bf16", A, M, B, N, C, N, M, N, T, 1, s)
torch.cuda.synchronize()
print("done")
