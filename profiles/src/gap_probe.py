"""Dependent-kernel boundary probe: the decode gate/up GEMV at 128 rows over 28 distinct weight buffers, timed with HIP events as an
eager launch loop and as a hipGraph replay, in a process that rocprofv3 --kernel-trace --stats can wrap: events (launch + boundary)
against the trace's kernel durations."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from video_r2_amd import lib
L = lib.load()
dev = torch.device('cuda:0')
M, I, H = 128, 18944, 3584
Wg = [torch.empty((2 * I // 16, H // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
st = torch.cuda.Stream()
s = st.cuda_stream


def launch_all():
    for W in Wg:
        lib.call("vr2_gemv_gateup_silu_bf16", x, H, M, W, I, H, act, I, s)


def ev():
    e = ctypes.c_void_p(); L.vr2_event_create(ctypes.byref(e)); return e


e0, e1 = ev(), ev()
with torch.cuda.stream(st):
    launch_all(); torch.cuda.synchronize()
    for mode in ("eager", "graph", "eager", "graph"):
        gex = None
        if mode == "graph":
            gex = ctypes.c_void_p()
            assert L.vr2_graph_begin(ctypes.c_void_p(s)) == 0
            launch_all()
            assert L.vr2_graph_end(ctypes.c_void_p(s), ctypes.byref(gex)) == 0
            L.vr2_graph_launch(gex, ctypes.c_void_p(s))
        torch.cuda.synchronize()
        L.vr2_event_record(e0, ctypes.c_void_p(s))
        for _ in range(10):
            if gex is not None:
                L.vr2_graph_launch(gex, ctypes.c_void_p(s))
            else:
                launch_all()
        L.vr2_event_record(e1, ctypes.c_void_p(s))
        torch.cuda.synchronize()
        ms = ctypes.c_float(); L.vr2_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        print(f"{mode}: {ms.value / 280 * 1e3:.2f} us per launch (HIP events over 280 launches)", flush=True)
