"""gate/up and q/k/v GEMMs at the training shape (T = 9280 rows) for PMC collection (L2 -> fabric traffic of the 256-row tile)."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
lib.load()
dev = torch.device('cuda:0')
s = torch.cuda.current_stream().cuda_stream
M, K = 9280, 3584
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
for N, epi in ((37888, 4), (4608, 1)):
    Ws = [torch.empty(((N + 15) // 16, K // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(4)]
    out = torch.empty(M, N // 2 if epi == 4 else N, dtype=torch.bfloat16 if epi == 4 else torch.float32, device=dev)
    for rep in range(3):
        for W in Ws:
            lib.call("vr2_gemm_bf16", x, K, W, None, out, out.stride(0), M, N, K, epi, s)
torch.cuda.synchronize()
print("done")
