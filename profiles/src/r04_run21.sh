#!/bin/bash
# round 4, run 21: the half-wave split of the ring GEMV (tune key 5: 0 = on, 1 = off): GEMV tests, then gate/up launch time + decode at P = 22
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "gemv or multi_group or decode_graph" > gpurun_out/r04_t21.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_t21.log
python - <<'PY'
import sys, torch, ctypes
sys.path.insert(0, '.')
from video_r2_amd import lib
L = lib.load()
dev = torch.device('cuda:0')
I, H = 18944, 3584
s = torch.cuda.current_stream().cuda_stream
for M in (176, 256, 160):
    Wg = [torch.empty((2 * I // 16, H // 32 * 512), dtype=torch.bfloat16, device=dev).normal_() for _ in range(28)]
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    outs = {}
    for key in (1, 0, 1, 0):
        L.vr2_tune_set(5, key)
        for W in Wg: lib.call("vr2_gemv_gateup_silu_bf16", x, H, M, W, I, H, act, I, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(5):
            for W in Wg: lib.call("vr2_gemv_gateup_silu_bf16", x, H, M, W, I, H, act, I, s)
        e1.record(); torch.cuda.synchronize()
        outs[key] = act.clone()
        print(f"M={M} half-split {'off' if key else 'on '}: {e0.elapsed_time(e1) / (5 * 28) * 1e3:.2f} us per launch", flush=True)
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    print(f"M={M} max |on - off| = {d:.3e} (max |out| {outs[1].float().abs().max().item():.2f})")
L.vr2_tune_set(5, 0)
PY
