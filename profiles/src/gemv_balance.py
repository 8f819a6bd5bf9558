import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from video_r2_amd import lib
L = lib.load(); dev = "cuda:0"; st = torch.cuda.Stream(device=dev); s = st.cuda_stream
def ev():
    e = ctypes.c_void_p(); L.vr2_event_create(ctypes.byref(e)); return e
def timed(fn, reps):
    e0, e1 = ev(), ev(); fn(); L.vr2_event_record(e0, ctypes.c_void_p(s))
    for _ in range(reps): fn()
    L.vr2_event_record(e1, ctypes.c_void_p(s)); st.synchronize()
    ms = ctypes.c_float(); L.vr2_event_elapsed_ms(e0, e1, ctypes.byref(ms)); return ms.value / reps
K = 3584; M = 128
with torch.cuda.stream(st):
    for I in (2048 * 8, 2048 * 8 + 64, 18944, 4096 * 8, 4096*8+64, 6144 * 8):
        Ws = [(torch.randn(2 * I * K // 8, 8, device=dev) * 0.05).to(torch.bfloat16) for _ in range(20)]
        x = torch.randn(M, K, device=dev).to(torch.bfloat16); out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        for v in (0, 1):
            L.vr2_tune_set(0, v)
            def run():
                for W in Ws: lib.call("vr2_gemv_gateup_silu_bf16", x, K, M, W, I, K, out, I, s)
            ms = timed(run, 3) / len(Ws)
            print(f"I={I} WGs={(2*I//16+7)//8} v{v}: {ms*1e3:.1f} us  {2*I*K*2/1e9/ms:.2f} TB/s", flush=True)
        del Ws; torch.cuda.empty_cache()
