"""Launch the decode gate/up GEMV (the roofline kernel) over 28 distinct layer-sized weight buffers, for PMC collection."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
lib.load()
dev = torch.device('cuda:0')
G, I, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 18944, 3584
Ws = [torch.empty((2 * I // 16, H // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
x = torch.randn(G, H, device=dev).to(torch.bfloat16)
act = torch.empty(G, I, dtype=torch.bfloat16, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for W in Ws:
        lib.call("vr2_gemv_gateup_silu_bf16", x, H, G, W, I, H, act, I, s)
torch.cuda.synchronize()
print("done")
