"""GRPO bookkeeping around the rollout: completion masks, reward aggregation, group advantages (GRPO + MAPO) and the
loss call.  Tiny tensors: torch on the host side for the [B,F] reward matrix, the HIP kernel ``vr2_grpo_loss`` for the
per-token loss/gradient.  Semantics of /root/reference/train/src/trainer/grpo_trainer.py:793-806, :907-981, :1099-1177.
"""
from __future__ import annotations

import torch

from .. import lib

LOSS_TYPES = {"grpo": 0, "bnpo": 1, "dr_grpo": 2, "gmpo": 3, "mapo": 4, "mapo_bnpo": 5}


def completion_mask(completion_ids, eos_token_id, mask_truncated=False):
    """-> (mask int32 [B,C] keeping tokens up to AND including the first EOS, eos_idx int64 [B], any_eos bool [B])."""
    is_eos = completion_ids == eos_token_id
    B, C = is_eos.shape
    any_eos = is_eos.any(dim=1)
    first = torch.where(any_eos, is_eos.int().argmax(dim=1), torch.full((B,), C, device=is_eos.device))
    cols = torch.arange(C, device=is_eos.device).expand(B, -1)
    mask = (cols <= first.unsqueeze(1)).int()
    if mask_truncated:
        mask = mask * any_eos.unsqueeze(1).int()
    return mask, first.long(), any_eos


def group_advantages(rewards_per_func, reward_weights, num_generations, loss_type="bnpo", scale_rewards=True,
                     reward_func_names=None, positive_reinforcement=None):
    """rewards_per_func fp32 [N,F] (NaN = function not applicable) -> dict(advantages, rewards, mean_grouped, std_grouped)."""
    r = rewards_per_func.clone().float()
    names = list(reward_func_names) if reward_func_names is not None else [f"f{i}" for i in range(r.shape[1])]
    G = num_generations
    if positive_reinforcement:
        wanted = {s.strip() for s in positive_reinforcement.split(",") if s.strip()} - {"accuracy_reward"}
        gate = (r[:, names.index("accuracy_reward")] > 0.1).float()
        for j, nm in enumerate(names):
            if nm in wanted:
                r[:, j] = r[:, j] * gate
    w = torch.as_tensor(reward_weights, dtype=torch.float32, device=r.device)
    total = (r * w.unsqueeze(0)).nansum(dim=1)
    grp = total.view(-1, G)
    mu = grp.mean(dim=1, keepdim=True)
    if loss_type in ("mapo", "mapo_bnpo"):
        sigma = grp.std(dim=1, keepdim=True, unbiased=False)
        eps = torch.finfo(grp.dtype).eps
        if "accuracy_reward" in names:
            wins = (r[:, names.index("accuracy_reward")].view(-1, G) >= 0.5).sum(dim=1, keepdim=True)
        else:
            wins = (grp >= mu).sum(dim=1, keepdim=True)
        p = wins.to(grp.dtype) / G
        lam = (1.0 - 4.0 * p * (1.0 - p)).clamp(0.0, 1.0)
        adv = (1.0 - lam) * ((grp - mu) / (sigma + eps)) + lam * ((grp - mu) / (mu + eps))
        std_g = sigma
    else:
        std_g = grp.std(dim=1, keepdim=True)
        adv = grp - mu
        if scale_rewards:
            adv = adv / (std_g + 1e-4)
    return dict(advantages=adv.reshape(-1), rewards=total, mean_grouped=mu.expand(-1, G).reshape(-1),
                std_grouped=std_g.expand(-1, G).reshape(-1), rewards_per_func=r)


def grpo_loss(logps, completion_mask_i32, advantages, *, ref_logps=None, old_logps=None, beta=0.04, epsilon_low=0.2,
              epsilon_high=0.2, loss_type="bnpo", max_completion_length=None, norm_rows=0, stream=0, want_grad=True):
    """Device tensors in, device tensors out: dict(loss, kl, clip_low, clip_high, clip_region, n_tokens [0-d], dlogp [B,C]).
    norm_rows: rows per normalisation block of bnpo / mapo_bnpo and of the kl / clip metrics (the reference's per-rank
    micro-batch, per_device_train_batch_size completions); 0 = the whole batch."""
    if loss_type not in LOSS_TYPES:
        raise ValueError(f"Unknown loss type: {loss_type}")
    B, C = logps.shape
    dev = logps.device
    out = torch.zeros(8, dtype=torch.float32, device=dev)
    grad = torch.empty((B, C), dtype=torch.float32, device=dev) if want_grad else None
    mcl = float(max_completion_length if max_completion_length is not None else C)
    lib.call("vr2_grpo_loss", logps.contiguous(), None if ref_logps is None else ref_logps.contiguous(),
             None if old_logps is None else old_logps.contiguous(), completion_mask_i32.to(torch.int32).contiguous(),
             advantages.to(torch.float32).contiguous(), B, C, float(beta), float(epsilon_low), float(epsilon_high),
             LOSS_TYPES[loss_type], mcl, int(norm_rows), out, grad, stream)
    return dict(loss=out[0], kl=out[1] if beta != 0.0 else None, clip_low=out[2], clip_high=out[3], clip_region=out[4],
                n_tokens=out[5], dlogp=grad)
