"""Data-parallel layer: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) or gloo on CPU.

Replaces the reference's accelerate + DeepSpeed ZeRO-3 stack for this path: weights are replicated
(7B bf16 + reference + fp32 optimizer state fit in 288 GB), prompt GROUPS are sharded over ranks so rewards and group
advantages need no collective (C1, C4-C7 eliminated), and the only exchanges per step are
  * one packed fp32 metrics all-reduce (C2/C8), and
  * one bucketed all-reduce (mean) of the trainable-gradient arena (C5 replacement): train.py lays the gradients out in
    backward-completion order and issues one bucket (a layer, ~0.93 GB fp32 at 7B) per backward stage, under the rest of the
    backward.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): the buckets are large so each collective runs bandwidth-bound on all
peers rather than latency-bound.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def collective_timeout_s():
    """Seconds a collective may stay incomplete before this rank gives up (VR2_COLLECTIVE_TIMEOUT, default 600): the process
    group's own timeout (RCCL's watchdog aborts the process when it expires) and the deadline of ExchangeWatch, which names the
    gradient bucket that is stuck."""
    return float(os.environ.get("VR2_COLLECTIVE_TIMEOUT", "600"))


def init(backend=None, device=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); no-op for 1 process."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return rank(), world_size()
    backend = backend or os.environ.get("VR2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"timeout": datetime.timedelta(seconds=collective_timeout_s())}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


class ExchangeWatch:
    """Deadline on the asynchronous gradient-bucket all-reduces.  The engine never blocks the host on a collective (the engine
    stream waits for RCCL's), so a rank whose peer is stuck or gone would sit in its next host sync until the backend's watchdog
    fires, with a message that names a sequence number.  A daemon thread polls the outstanding works instead: a bucket that has
    not completed `timeout` seconds after it was issued ends THIS process with exit code 17 and one line on stderr that names the
    bucket, its size and the rank -- the launcher (torch.distributed.run) then tears the job down.  Nothing is re-executed or
    retried: a gradient exchange that did not happen cannot be repaired inside the step."""
    EXIT_CODE = 17

    def __init__(self, timeout=None, poll=0.25, on_timeout=None):
        import threading
        self.timeout = collective_timeout_s() if timeout is None else float(timeout)
        self.poll = poll
        self.on_timeout = on_timeout                 # tests: called instead of os._exit
        self._lock = threading.Lock()
        self._items = []                              # (name, work, nbytes, issued_at)
        self._thread = None
        self._stop = False

    def add(self, name, work, nbytes):
        import threading
        import time
        with self._lock:
            self._items.append((str(name), work, int(nbytes), time.monotonic()))
            if self._thread is None:
                self._thread = threading.Thread(target=self._run, name="vr2-exchange-watch", daemon=True)
                self._thread.start()

    def clear(self):
        """the step has consumed the exchanged gradients: every registered work is known to be complete"""
        with self._lock:
            self._items = []

    def close(self):
        self._stop = True

    def _run(self):
        import sys
        import time
        while not self._stop:
            time.sleep(self.poll)
            now = time.monotonic()
            with self._lock:
                try:
                    self._items = [it for it in self._items if not it[1].is_completed()]
                except Exception:                    # the process group went away under the poll (shutdown): nothing left to watch
                    self._items = []
                late = [it for it in self._items if now - it[3] > self.timeout]
            if late:
                name, _, nbytes, t0 = late[0]
                msg = (f"[video_r2_amd.dist] rank {rank()} of {world_size()}: the all-reduce of gradient bucket '{name}' ({nbytes / 2 ** 20:.0f} MiB) "
                       f"has not completed {now - t0:.0f} s after it was issued (VR2_COLLECTIVE_TIMEOUT={self.timeout:g}): a peer rank is stuck "
                       f"or gone.  Exiting with code {self.EXIT_CODE}; {len(self._items)} bucket(s) outstanding.")
                print(msg, file=sys.stderr, flush=True)
                if self.on_timeout is not None:
                    self.on_timeout(name)
                    return
                os._exit(self.EXIT_CODE)


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_prompts(prompt_indices, r=None, w=None):
    """The step's prompt groups dealt contiguously to ranks (every rank computed the same seeded order)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = len(prompt_indices) // w
    return list(prompt_indices[r * per:(r + 1) * per])


class MetricsReducer:
    """Named scalars -> ONE fp32 all-reduce per step.  kinds: mean (default), sum, min, max (min/max ride in the same
    vector through a second and third reduction only when such keys exist)."""

    def __init__(self, device="cpu"):
        self.device = device

    def reduce(self, values: dict, kinds: dict | None = None):
        kinds = kinds or {}
        names = sorted(values)
        out = {}
        for op_name, op in (("mean", dist.ReduceOp.SUM), ("sum", dist.ReduceOp.SUM), ("min", dist.ReduceOp.MIN), ("max", dist.ReduceOp.MAX)):
            sel = [n for n in names if kinds.get(n, "mean") == op_name]
            if not sel:
                continue
            v = torch.tensor([float(values[n]) for n in sel], dtype=torch.float32, device=self.device)
            # NaN-aware mean as gather_for_metrics(...).nanmean() in the reference: carry a validity count
            if op_name == "mean":
                ok = ~torch.isnan(v)
                v = torch.cat([torch.where(ok, v, torch.zeros_like(v)), ok.float()])
            if dist.is_initialized() and world_size() > 1:
                dist.all_reduce(v, op=op)
            if op_name == "mean":
                n = len(sel)
                tot, cnt = v[:n], v[n:]
                v = torch.where(cnt > 0, tot / cnt.clamp(min=1), torch.full_like(tot, float("nan")))
            for n_, x in zip(sel, v.tolist()):
                out[n_] = x
        return out


def allreduce_mean_(flat: torch.Tensor, bucket_bytes=512 << 20):
    """In-place mean all-reduce of a flat gradient buffer in large buckets (async, waited at the end)."""
    w = world_size()
    if w == 1:
        return flat
    n = flat.numel()
    per = max(1, bucket_bytes // flat.element_size())
    works = []
    for a in range(0, n, per):
        works.append(dist.all_reduce(flat[a:a + per], op=dist.ReduceOp.SUM, async_op=True))
    for wk in works:
        wk.wait()
    flat.div_(w)
    return flat
