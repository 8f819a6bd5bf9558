"""vr2_gemm_a8w8 (block-scaled fp8 MFMA) against vr2_gemm_bf16 on the LM shapes of the no-grad passes, HIP events, random operands.
    python profiles/src/gemm_fp8_bench.py [rows]        (default 8192 = G x C scoring rows of one prompt group at the bench shape)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_r2_amd import lib
from video_r2_amd.weights import pack_tiles, pack_tiles_fp8, quantize_rows_fp8

lib.load()
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = lambda: torch.cuda.current_stream().cuda_stream
H, I, V = 3584, 18944, 152064
shapes = [("q/k/v", 4608, H, 1), ("o (+=)", H, H, 2), ("gate/up + SwiGLU", 2 * I, H, 4), ("down (+=)", H, I, 2), ("lm_head (4096 rows)", V, H, 1)]
for name, N, K, epi in shapes:
    M = 4096 if name.startswith("lm_head") else T
    g = torch.Generator(device="cuda").manual_seed(1)
    Wf = (0.02 * torch.randn(N, K, device=dev, generator=g))
    Wb = pack_tiles(Wf.to(torch.bfloat16))
    q, s = quantize_rows_fp8(Wf)
    W8 = pack_tiles_fp8(q)
    del Wf, q
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    q8 = torch.empty((M, K), dtype=torch.uint8, device=dev)
    xs = torch.empty(M, device=dev)
    lib.call("vr2_quant_rows_fp8", x, K, M, K, q8, K, xs, S())
    No = N // 2 if epi == 4 else N
    C = torch.zeros((M, No), dtype=torch.bfloat16 if epi == 4 else torch.float32, device=dev)
    def run_b():
        lib.call("vr2_gemm_bf16", x, K, Wb, None, C, No, M, N, K, epi, S())
    def run_8():
        lib.call("vr2_gemm_a8w8", q8, K, xs, W8, s, None, C, No, M, N, K, epi, S())
    def run_q():
        lib.call("vr2_quant_rows_fp8", x, K, M, K, q8, K, xs, S())
    res = {}
    for tag, fn in (("bf16", run_b), ("fp8", run_8), ("quant", run_q)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * M * N * K
    print(f"{name:22s} {M} x {N} x {K}: bf16 {res['bf16']:8.1f} us = {fl / res['bf16'] / 1e6:7.1f} TFLOP/s | fp8 {res['fp8']:8.1f} us = "
          f"{fl / res['fp8'] / 1e6:7.1f} TFLOP/s ({res['bf16'] / res['fp8']:.2f}x) | row quantiser of the input {res['quant']:6.1f} us", flush=True)
    del Wb, W8, x, q8, C
    torch.cuda.empty_cache()
