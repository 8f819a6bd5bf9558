"""GRPO algebra of the reference trainer, restated in numpy (float32 arithmetic, explicit loops).

Test infrastructure only (see oracle/__init__.py).
Follows /root/reference/train/src/trainer/grpo_trainer.py: reward aggregation + advantages :907-981,
loss :1085-1178, nanstd/nanmin/nanmax :180-253.  The reference trainer itself is not importable here
(trl / deepspeed / peft absent) and ships no tests: beyond the hand-computed known answers in
tests/test_host_cpu.py (advantages, loss vs torch autograd, the 8-ranks-x-1-completion emulation) this file is "parity unpinned".
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
EPS_F32 = np.finfo(np.float32).eps


def nansum_rows(x):
    out = np.zeros((x.shape[0],), dtype=F32)
    for i in range(x.shape[0]):
        acc = F32(0)
        for v in x[i]:
            if not math.isnan(v):
                acc = F32(acc + F32(v))
        out[i] = acc
    return out


def nanstd(x):
    """grpo_trainer.py:180-195 (Bessel-corrected, NaNs ignored)."""
    x = np.asarray(x, dtype=F32)
    ok = x[~np.isnan(x)]
    n = ok.size
    if n == 0:
        return F32(np.nan)
    m = ok.mean(dtype=F32)
    var = ((ok - m) ** 2).mean(dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        var = F32(var * F32(n) / F32(n - 1))
    return F32(np.sqrt(var))


def nanmin(x):
    x = np.asarray(x, dtype=F32)
    return F32(np.nan) if np.isnan(x).all() else x[~np.isnan(x)].min()


def nanmax(x):
    x = np.asarray(x, dtype=F32)
    return F32(np.nan) if np.isnan(x).all() else x[~np.isnan(x)].max()


def advantages(rewards_per_func, reward_weights, num_generations, loss_type="bnpo", scale_rewards=True,
               reward_func_names=None, positive_reinforcement=None):
    """grpo_trainer.py:909-974.  rewards_per_func [N,F] fp32 with NaN for 'not applicable' -> dict with
    advantages [N], rewards [N], mean_grouped [N], std_grouped [N]."""
    r = np.array(rewards_per_func, dtype=F32, copy=True)
    names = list(reward_func_names) if reward_func_names is not None else [f"f{i}" for i in range(r.shape[1])]
    G = num_generations
    if positive_reinforcement:                                          # :910-917
        pos = [s.strip() for s in positive_reinforcement.split(",") if s.strip()]
        acc_idx = names.index("accuracy_reward")
        m = (r[:, acc_idx] > F32(0.1)).astype(F32)
        for j, nm in enumerate(names):
            if nm in pos and nm != "accuracy_reward":
                r[:, j] = r[:, j] * m
    wts = np.asarray(reward_weights, dtype=F32)
    rewards = nansum_rows(r * wts[None, :])                              # :920
    N = rewards.shape[0]
    adv = np.zeros((N,), dtype=F32)
    mean_g = np.zeros((N,), dtype=F32)
    std_g = np.zeros((N,), dtype=F32)
    for g0 in range(0, N, G):
        grp = rewards[g0:g0 + G]
        mu = grp.mean(dtype=F32)
        if loss_type in ("mapo", "mapo_bnpo"):                           # :922-962
            sigma = F32(np.sqrt(((grp - mu) ** 2).mean(dtype=F32)))      # biased
            if "accuracy_reward" in names:
                acc = r[g0:g0 + G, names.index("accuracy_reward")]
                succ = int((acc >= F32(0.5)).sum())
            else:
                succ = int((grp >= mu).sum())
            p = F32(succ) / F32(G)
            lam = F32(1.0) - F32(4.0) * p * (F32(1.0) - p)
            lam = min(max(lam, F32(0.0)), F32(1.0))
            z = (grp - mu) / (sigma + EPS_F32)
            apd = (grp - mu) / (mu + EPS_F32)
            adv[g0:g0 + G] = (F32(1.0) - lam) * z + lam * apd
            mean_g[g0:g0 + G] = mu
            std_g[g0:g0 + G] = sigma
        else:                                                            # :964-974
            if G > 1:
                sd = F32(np.sqrt(((grp - mu) ** 2).sum(dtype=F32) / F32(G - 1)))   # unbiased
            else:
                sd = F32(np.nan)
            a = grp - mu
            if scale_rewards:
                a = a / (sd + F32(1e-4))
            adv[g0:g0 + G] = a
            mean_g[g0:g0 + G] = mu
            std_g[g0:g0 + G] = sd
    return dict(advantages=adv, rewards=rewards, mean_grouped=mean_g, std_grouped=std_g, rewards_per_func=r)


def loss(per_token_logps, completion_mask, advantages, *, ref_per_token_logps=None, old_per_token_logps=None,
         beta=0.04, epsilon_low=0.2, epsilon_high=0.2, loss_type="bnpo", max_completion_length=None, norm_rows=0):
    """grpo_trainer.py:1099-1177 in float64-free float32 numpy, for ONE rank's micro-batch (norm_rows = 0), or -- norm_rows =
    u > 0 -- for the rows cut into micro-batches of u consecutive rows whose losses / metrics are then averaged, which is what
    the reference's ranks and accumulation steps compute together (each rank: :1146-1177 on per_device_train_batch_size rows;
    DDP and Trainer average the ranks' and steps' losses; gather_for_metrics(...).nanmean() the metrics).

    Returns dict(loss, per_token_loss [B,C], dloss_dlogp [B,C] (analytic gradient wrt per_token_logps with
    old = logp.detach() when old is None), kl, clip_low, clip_high, clip_region).
    """
    if norm_rows and 0 < norm_rows < np.asarray(per_token_logps).shape[0]:
        u, Bt = int(norm_rows), np.asarray(per_token_logps).shape[0]
        parts = []
        for b0 in range(0, Bt, u):
            sl = slice(b0, b0 + u)
            parts.append(loss(np.asarray(per_token_logps)[sl], np.asarray(completion_mask)[sl], np.asarray(advantages)[sl],
                              ref_per_token_logps=None if ref_per_token_logps is None else np.asarray(ref_per_token_logps)[sl],
                              old_per_token_logps=None if old_per_token_logps is None else np.asarray(old_per_token_logps)[sl],
                              beta=beta, epsilon_low=epsilon_low, epsilon_high=epsilon_high, loss_type=loss_type,
                              max_completion_length=max_completion_length))
        n = F32(len(parts))
        with np.errstate(invalid="ignore"):
            mean = lambda k: (None if parts[0][k] is None else F32(np.nanmean(np.asarray([p[k] for p in parts], dtype=F32))))  # noqa: E731
        return dict(loss=F32(sum(p["loss"] for p in parts) / n), per_token_loss=np.concatenate([p["per_token_loss"] for p in parts]),
                    dloss_dlogp=np.concatenate([p["dloss_dlogp"] for p in parts]) / n, kl=mean("kl"), clip_low=mean("clip_low"),
                    clip_high=mean("clip_high"), clip_region=mean("clip_region"))
    lp = np.asarray(per_token_logps, dtype=F32)
    mask = np.asarray(completion_mask).astype(F32)
    A = np.asarray(advantages, dtype=F32)[:, None]
    B, C = lp.shape
    old = lp if old_per_token_logps is None else np.asarray(old_per_token_logps, dtype=F32)
    if beta != 0.0:
        ref = np.asarray(ref_per_token_logps, dtype=F32)
        d = ref - lp
        kl = np.exp(d) - d - F32(1)                                      # :1101-1103
        dkl = -np.exp(d) + F32(1)                                        # d kl / d lp
    else:
        kl = np.zeros_like(lp)
        dkl = np.zeros_like(lp)
    cnt = np.maximum(mask.sum(-1), F32(1))
    if loss_type == "gmpo":                                              # :1119-1126
        delta = lp - old
        seq = (delta * mask).sum(-1) / cnt
        coef1 = np.repeat(np.exp(seq)[:, None], C, axis=1)
        dcoef1 = coef1 * (mask / cnt[:, None])     # d coef1[b,*] / d lp[b,t]  (same for every column of row b)
    else:
        coef1 = np.exp(lp - old)                                         # :1129
        dcoef1 = coef1
    lo, hi = F32(1 - epsilon_low), F32(1 + epsilon_high)
    coef2 = np.clip(coef1, lo, hi)                                       # :1132
    l1 = coef1 * A
    l2 = coef2 * A
    ptl = -np.minimum(l1, l2)                                            # :1135-1137
    # gradient of -min(l1,l2) wrt coef1: -A where l1 is the (non-strict) min branch... torch.min backward
    # sends the gradient to the smaller operand and splits it evenly on ties; clamp passes gradient inside [lo,hi].
    inside = ((coef1 >= lo) & (coef1 <= hi)).astype(F32)
    g1 = np.where(l1 < l2, F32(1), np.where(l1 == l2, F32(0.5), F32(0)))
    g2 = np.where(l2 < l1, F32(1), np.where(l1 == l2, F32(0.5), F32(0)))
    dptl_dcoef1 = -(g1 * A + g2 * A * inside)
    if beta != 0.0:
        ptl = ptl + F32(beta) * kl                                       # :1138-1139
    # aggregation :1146-1153 ; w[b,t] = d loss / d ptl[b,t]
    if loss_type in ("grpo", "gmpo", "mapo"):
        val = ((ptl * mask).sum(-1) / cnt).mean(dtype=F32)
        w = mask / cnt[:, None] / F32(B)
    elif loss_type in ("bnpo", "mapo_bnpo"):
        tot = max(mask.sum(dtype=F32), F32(1))
        val = (ptl * mask).sum(dtype=F32) / tot
        w = mask / tot
    elif loss_type == "dr_grpo":
        den = F32(B * max_completion_length)
        val = (ptl * mask).sum(dtype=F32) / den
        w = mask / den
    else:
        raise ValueError(f"Unknown loss type: {loss_type}")
    if loss_type == "gmpo":
        # every column of row b depends on every lp[b,t]: chain through the sequence ratio
        s = (w * dptl_dcoef1).sum(-1, keepdims=True)
        grad = s * dcoef1 + w * F32(beta) * dkl
    else:
        grad = w * (dptl_dcoef1 * dcoef1 + F32(beta) * dkl)
    if old_per_token_logps is not None and loss_type != "gmpo":
        pass  # old is a constant either way; formula above already treats it as detached
    msum = mask.sum(dtype=F32)
    low = ((coef1 < lo) & (A < 0)).astype(F32)
    high = ((coef1 > hi) & (A > 0)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = dict(loss=F32(val), per_token_loss=ptl, dloss_dlogp=grad.astype(F32),
                   kl=F32((kl * mask).sum(dtype=F32) / msum) if beta != 0.0 else None,
                   clip_low=F32((low * mask).sum(dtype=F32) / msum),
                   clip_high=F32((high * mask).sum(dtype=F32) / msum),
                   clip_region=F32((np.maximum(low, high) * mask).sum(dtype=F32) / msum))
    return out


def completion_metrics(completion_mask, any_eos):
    """grpo_trainer.py:989-1004."""
    lens = np.asarray(completion_mask).sum(1).astype(F32)
    term = lens[np.asarray(any_eos, dtype=bool)]
    clipped = 1 - len(term) / len(lens)
    if len(term) == 0:
        term = np.zeros((1,), dtype=F32)
    return {"completions/mean_length": float(lens.mean()), "completions/min_length": float(lens.min()),
            "completions/max_length": float(lens.max()), "completions/clipped_ratio": float(clipped),
            "completions/mean_terminated_length": float(term.mean()),
            "completions/min_terminated_length": float(term.min()),
            "completions/max_terminated_length": float(term.max())}
