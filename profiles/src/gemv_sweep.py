"""Decode GEMV A/B sweep on one MI355X: every 7B decode projection at 32 / 64 / 128 rows, kernel variants selected with
vr2_tune_set(0, v) (0 = LDS-staged kernel of round 1, 1 = register-ring kernel, 2 = ring with one chunk in flight).
HIP events on one stream over 28 distinct weight buffers per shape (nothing is cache-resident), results checked against
variant 0.  Usage: python profiles/src/gemv_sweep.py [rows ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_r2_amd import lib  # noqa: E402

L = lib.load()
dev = "cuda:0"
st = torch.cuda.Stream(device=dev)
s = st.cuda_stream


def ev():
    e = ctypes.c_void_p()
    L.vr2_event_create(ctypes.byref(e))
    return e


def timed(fn, reps):
    e0, e1 = ev(), ev()
    fn()
    L.vr2_event_record(e0, ctypes.c_void_p(s))
    for _ in range(reps):
        fn()
    L.vr2_event_record(e1, ctypes.c_void_p(s))
    st.synchronize()
    ms = ctypes.c_float()
    L.vr2_event_elapsed_ms(e0, e1, ctypes.byref(ms))
    return ms.value / reps


H, I, V = 3584, 18944, 152064
shapes = [("gate/up", 1, I, H, 1), ("qkv", 0, 4608, H, 7), ("o", 0, H, H, 7), ("down", 0, H, I, 8), ("lm_head", 0, V, H, 1)]
rows = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
NL = 28
with torch.cuda.stream(st):
    for name, mode, N, K, nsplit in shapes:
        nl = 2 if name == "lm_head" else NL
        n_w = (2 * N if mode == 1 else N)
        Ws = [(torch.randn(n_w * K // 8, 8, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nl)]   # packed layout: any bytes do
        for M in rows:
            x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
            if mode == 1:
                out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            else:
                out = torch.empty(nsplit, M, N, dtype=torch.float32, device=dev)
            res = {}
            line = f"{name:8s} M={M:3d} N={N} K={K} split={nsplit}: "
            for v in (0, 1, 2):
                L.vr2_tune_set(0, v)

                def run():
                    for W in Ws:
                        if mode == 1:
                            lib.call("vr2_gemv_gateup_silu_bf16", x, K, M, W, N, K, out, N, s)
                        else:
                            lib.call("vr2_gemv_bf16", x, K, M, W, N, K, out, out.stride(0) if nsplit > 1 else 0, N, nsplit, s)
                ms = timed(run, 3) / nl
                st.synchronize()
                r = out.float().sum(0) if mode == 0 else out.float()
                res[v] = r.clone()
                gb = n_w * K * 2 / 1e9
                line += f" v{v} {ms * 1e3:7.1f} us {gb / ms:5.2f} TB/s |"
            if mode == 1 and M == 128:       # what bounds the ring kernel: time without x staging / LDS+MFMA / weight stream
                L.vr2_tune_set(0, 2)
                for abl, what in ((1, "no x staging"), (2, "no LDS reads + MFMA"), (3, "no weight stream"), (4, "no x, no weight stream")):
                    L.vr2_tune_set(1, abl)
                    ms = timed(run, 3) / nl
                    line += f" [{what} {ms * 1e3:.1f} us]"
                L.vr2_tune_set(1, 0)
            if True:                          # x handed over chunk-major ([K/64][M][64]) instead of row-major
                L.vr2_tune_set(0, 1)
                L.vr2_tune_set(2, 1)
                xs = x
                x = x.view(M, K // 64, 64).permute(1, 0, 2).contiguous()
                ms = timed(run, 3) / nl
                st.synchronize()
                rcm = (out.float().sum(0) if mode == 0 else out.float()).clone()
                line += f" [chunk-major x {ms * 1e3:.1f} us, diff {(rcm - res[1]).abs().max().item():.2g}]"
                if mode == 1 and M == 128:
                    L.vr2_tune_set(1, 3)
                    ms = timed(run, 3) / nl
                    line += f" [chunk-major, no weight stream {ms * 1e3:.1f} us]"
                    L.vr2_tune_set(1, 0)
                x = xs
                L.vr2_tune_set(2, 0)
            d1 = (res[1] - res[0]).abs().max().item()
            d2 = (res[2] - res[0]).abs().max().item()
            ref = res[0].abs().max().item()
            print(line + f" maxdiff v1 {d1:.3g} v2 {d2:.3g} (scale {ref:.3g})", flush=True)
        del Ws
        torch.cuda.empty_cache()
