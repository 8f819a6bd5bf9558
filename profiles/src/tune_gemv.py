import sys, ctypes, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
L = lib.load()
dev = torch.device('cuda:0')
st = torch.cuda.Stream()
s = st.cuda_stream
def ev():
    e = ctypes.c_void_p(); L.vr2_event_create(ctypes.byref(e)); return e
def timed_graph(fn, reps=5):
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        lib.call("vr2_graph_begin", s)
        fn()
        ex = ctypes.c_void_p(); L.vr2_graph_end(ctypes.c_void_p(s), ctypes.byref(ex))
        lib.call("vr2_graph_launch", ex, s); st.synchronize()
        e0, e1 = ev(), ev()
        L.vr2_event_record(e0, ctypes.c_void_p(s))
        for _ in range(reps): lib.call("vr2_graph_launch", ex, s)
        L.vr2_event_record(e1, ctypes.c_void_p(s)); st.synchronize()
        ms = ctypes.c_float(); L.vr2_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps
G = 8
NL = 28
for (N, K, name) in [(4608, 3584, "qkv"), (3584, 3584, "o"), (3584, 18944, "down"), (152064, 3584, "lm_head")]:
    nl = NL if N < 100000 else 4
    Ws = [torch.empty(((N + 15) // 16, K // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(nl)]
    x = torch.randn(G, K, device=dev).to(torch.bfloat16)
    for ns in (1, 2, 4, 8):
        if (K // 32) < ns * 4: continue
        out = torch.empty(ns, G, N, device=dev)
        def fn():
            for W in Ws:
                lib.call("vr2_gemv_bf16", x, K, G, W, N, K, out, G * N, N, ns, s)
        ms = timed_graph(fn)
        us = ms * 1e3 / nl
        print(f"{name:8s} N={N:6d} K={K:5d} nsplit={ns}: {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s")
# small kernels
H = 3584
x = torch.randn(G, H, device=dev); w = torch.ones(H, device=dev); y = torch.empty(G, H, dtype=torch.bfloat16, device=dev)
for ns in (1, 2, 4):
    part = torch.randn(ns, G, H, device=dev)
    def fn():
        for _ in range(28):
            lib.call("vr2_add_rmsnorm", x, H, part, G * H, H, ns, None, w, y, H, G, H, 1e-6, s)
    print(f"add_rmsnorm ns={ns}: {timed_graph(fn)*1e3/28:.2f} us")
def fn():
    for _ in range(28):
        lib.call("vr2_add_int", y, 0, s)
print(f"empty-ish kernel (add_int): {timed_graph(fn)*1e3/28:.2f} us")
