"""Launch the decode gate/up GEMV and the split-K down projection at M rows over 28 distinct layer-sized weight buffers, for PMC
collection (round 2: gemv_ring_kernel)."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
lib.load()
dev = torch.device('cuda:0')
M, I, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 18944, 3584
NS = 8
Wg = [torch.empty((2 * I // 16, H // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
Wd = [torch.empty((H // 16, I // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
pd = torch.empty(NS, M, H, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for W in Wg:
        lib.call("vr2_gemv_gateup_silu_bf16", x, H, M, W, I, H, act, I, s)
    for W in Wd:
        lib.call("vr2_gemv_bf16", act, I, M, W, H, I, pd, pd.stride(0), H, NS, s)
torch.cuda.synchronize()
print("done")
