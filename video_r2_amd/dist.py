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


def init(backend=None, device=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); no-op for 1 process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return rank(), world_size()
    backend = backend or os.environ.get("VR2_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


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
