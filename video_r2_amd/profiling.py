"""Phase timing of the trainer: what the reference's TRL helpers do for its trainer (``@profiling_decorator`` on
``_get_per_token_logps`` / ``compute_loss`` and ``profiling_context(self, reward_name)`` around every reward function,
/root/reference/train/src/trainer/grpo_trainer.py:46,711,722,850,1066: wall time logged as ``profiling/Time taken: <Class>.<name>``),
done the way this engine needs it:

* device phases are bracketed by HIP events on the ENGINE's stream (the host only enqueues: its wall clock says nothing about a
  phase) and read once per step, after the step's one host sync;
* every phase is also a roctx range (``libroctx64``), so ``rocprofv3 --marker-trace`` shows the same names on the timeline;
* host phases (the reward functions, which run on their own thread) are plain wall-clock intervals.

The metric names keep the reference's form, so dashboards built on its logs keep working.
"""
from __future__ import annotations

import contextlib
import ctypes
import time
from collections import defaultdict

import torch

_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
        else:
            _roctx = False
    return _roctx


class PhaseTimer:
    """with timer.phase("name"): ... (device work enqueued on ``stream``) / with timer.host("name"): ... (host work);
    collect() after a host sync -> {"profiling/Time taken: <owner>.<name>": seconds} summed over the step, then resets."""

    def __init__(self, owner="QwenGRPOTrainer", stream=None, enabled=True):
        self.owner = owner
        self.stream = stream
        self.enabled = enabled
        self._events = []                       # (name, start event, end event)
        self._host = defaultdict(float)

    @contextlib.contextmanager
    def phase(self, name):
        if not self.enabled:
            yield
            return
        rt = _load_roctx()
        if rt:
            rt.roctxRangePushA(f"{self.owner}.{name}".encode())
        ev = None
        if self.stream is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(self.stream)
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if ev is not None:
                ev[1].record(self.stream)
                self._events.append((name, ev[0], ev[1]))
            else:
                self._host[name] += time.perf_counter() - t0
            if rt:
                rt.roctxRangePop()

    @contextlib.contextmanager
    def host(self, name):
        if not self.enabled:
            yield
            return
        rt = _load_roctx()
        if rt:
            rt.roctxRangePushA(f"{self.owner}.{name}".encode())
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self._host[name] += time.perf_counter() - t0
            if rt:
                rt.roctxRangePop()

    def collect(self, expect=()):
        """expect: names reported even when nothing ran under them this step (0.0) -- every rank must hand the same keys to the
        packed metrics all-reduce"""
        out = defaultdict(float)
        for name in expect:
            out[name] += 0.0
        for name, e0, e1 in self._events:
            out[name] += e0.elapsed_time(e1) * 1e-3          # (the caller has synchronised the stream)
        for name, t in self._host.items():
            out[name] += t
        self._events, self._host = [], defaultdict(float)
        return {f"profiling/Time taken: {self.owner}.{k}": v for k, v in out.items()}
