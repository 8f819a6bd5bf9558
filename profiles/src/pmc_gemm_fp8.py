"""Driver for the rocprofv3 passes over vr2_gemm_a8w8 next to vr2_gemm_bf16 (profiles/src/r03_profiles.sh): the gate/up + SwiGLU
and down-projection shapes at 8192 rows, 3 launches each (the first = first touch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_r2_amd import lib
from video_r2_amd.weights import pack_tiles, pack_tiles_fp8, quantize_rows_fp8

lib.load()
dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream
T, H, I = 8192, 3584, 18944
for name, N, K, epi in [("gate/up", 2 * I, H, 4), ("down", H, I, 2)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    Wf = 0.02 * torch.randn(N, K, device=dev, generator=g)
    Wb = pack_tiles(Wf.to(torch.bfloat16))
    q, s = quantize_rows_fp8(Wf)
    W8 = pack_tiles_fp8(q)
    del Wf, q
    x = torch.randn(T, K, device=dev, generator=g).to(torch.bfloat16)
    q8 = torch.empty((T, K), dtype=torch.uint8, device=dev)
    xs = torch.empty(T, device=dev)
    lib.call("vr2_quant_rows_fp8", x, K, T, K, q8, K, xs, S())
    No = N // 2 if epi == 4 else N
    C = torch.zeros((T, No), dtype=torch.bfloat16 if epi == 4 else torch.float32, device=dev)
    for _ in range(3):
        lib.call("vr2_gemm_bf16", x, K, Wb, None, C, No, T, N, K, epi, S())
    for _ in range(3):
        lib.call("vr2_gemm_a8w8", q8, K, xs, W8, s, None, C, No, T, N, K, epi, S())
    torch.cuda.synchronize()
    del Wb, W8, x, q8, C
    torch.cuda.empty_cache()
