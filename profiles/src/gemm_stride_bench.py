"""Does the row stride of the row-major GEMM operands matter (L2 channel spread)?  Same shapes, leading dimensions padded."""
import sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib, weights as W
lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
T = 9280

def timeit(name, fn, flops, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:60s} {ms*1e3:9.1f} us   {flops/ms/1e9:8.1f} TFLOP/s", flush=True)

def strided(rows, cols, pad):
    buf = torch.randn(rows, cols + pad, device=dev, generator=g).to(torch.bfloat16)
    return buf[:, :cols]

# wgrad: C[M][N] += A[T][M]^T B[T][N]
for (M, N) in ((37888, 3584), (3584, 18944), (4608, 3584)):
    c = torch.zeros(M, N, device=dev)
    for pa, pb in ((0, 0), (128, 0), (0, 128), (128, 128), (64, 64), (32, 32)):
        a = strided(T, M, pa); b_ = strided(T, N, pb)
        timeit(f"wgrad {M}x{N} lda+{pa} ldb+{pb}", lambda: lib.call("vr2_gemm_tn_bf16", a, a.stride(0), b_, b_.stride(0), c, N, M, N, T, 1, st), 2.0 * T * M * N)
        del a, b_
    del c
# NT: C[T][N] = A[T][K] W[N][K]^T (W tile-packed)
for (N, K, epi) in ((4608, 3584, 1), (3584, 3584, 2), (3584, 18944, 2), (18944, 3584, 0)):
    w = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    wp = W.pack_tiles(w)
    for pa in (0, 128, 64, 32):
        a = strided(T, K, pa)
        c = torch.zeros(T, N, device=dev, dtype=torch.float32 if epi else torch.bfloat16)
        timeit(f"gemm NT T x {N} x {K} epi {epi} lda+{pa}", lambda: lib.call("vr2_gemm_bf16", a, a.stride(0), wp, None, c, N, T, N, K, epi, st), 2.0 * T * N * K)
        del a, c
    del w, wp
