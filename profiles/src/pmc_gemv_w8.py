"""Launch the fp8-weight decode gate/up GEMV over 28 distinct layer-sized weight buffers, for PMC collection (config 5)."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
lib.load()
dev = torch.device('cuda:0')
G, I, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 18944, 3584
Ws = [torch.randint(0, 120, (2 * I // 16, H // 64 * 1024), dtype=torch.uint8, device=dev) for _ in range(28)]
sc = torch.ones(2 * I, device=dev)
x = torch.randn(G, H, device=dev).to(torch.bfloat16)
act = torch.empty(G, I, dtype=torch.bfloat16, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for W in Ws:
        lib.call("vr2_gemv_gateup_silu_w8", x, H, G, W, sc, I, H, act, I, s)
torch.cuda.synchronize()
print("done")
