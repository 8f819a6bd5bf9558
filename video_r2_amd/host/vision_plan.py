"""Index plans for the vision tower, computed once per video geometry with vectorised numpy (bit-exact integers).

What the reference reaches through HF ``transformers.vision_utils`` (get_vision_window_index :130-188,
get_vision_cu_seqlens :42-65, get_vision_position_ids :81-127; reference call site
/root/reference/train/src/trainer/grpo_trainer.py:725-727).  Also builds the attention work items the
``vr2_attn_tok`` kernel consumes (include/vr2.h).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

Q_TILE = 64  # query rows per attention workgroup, head_dim 80 (the ViT: 4 waves x 16 slots)
LM_Q_TILE = 128  # head_dim 128 (the LM forward and dQ kernels: 4 waves x 32 rows)


def window_index(grid_thw, merge_size=2, window_size=112, patch_size=14):
    """-> (window_index int64 [sum t*h*w/merge^2], cu_window_seqlens int32, deduplicated)."""
    win = window_size // merge_size // patch_size
    unit = merge_size * merge_size
    idx_parts, cu, base = [], [np.zeros(1, dtype=np.int64)], 0
    last = 0
    for t, h, w in np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3):
        lh, lw = int(h) // merge_size, int(w) // merge_size
        ph, pw = win - lh % win, win - lw % win
        nh, nw = (lh + ph) // win, (lw + pw) // win
        idx = np.arange(t * lh * lw, dtype=np.int64).reshape(t, lh, lw)
        padded = np.full((t, lh + ph, lw + pw), -100, dtype=np.int64)
        padded[:, :lh, :lw] = idx
        padded = padded.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, win * win)
        valid = padded != -100
        idx_parts.append(padded[valid] + base)
        lens = valid.sum(-1).reshape(-1) * unit
        c = np.cumsum(lens) + last
        cu.append(c)
        last = int(c[-1])
        base += int(t) * lh * lw
    cu = np.concatenate(cu)
    keep = np.ones(cu.shape, dtype=bool)
    keep[1:] = cu[1:] != cu[:-1]
    return np.concatenate(idx_parts), cu[keep].astype(np.int32)


def full_cu_seqlens(grid_thw):
    g = np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3)
    lens = np.repeat(g[:, 1] * g[:, 2], g[:, 0])
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def position_ids_hw(grid_thw, merge_size=2):
    out = []
    for t, h, w in np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3):
        hh, ww = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shp = (h // merge_size, merge_size, w // merge_size, merge_size)
        hh = hh.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hh, ww], -1), (int(t), 1)))
    return np.concatenate(out).astype(np.int64)


def segment_work(cu_seqlens, q_tile=Q_TILE):
    """Non-causal varlen segments -> vr2_attn_tok work items int32 [n,8]: every q tile sees its whole segment."""
    items = []
    cu = np.asarray(cu_seqlens, dtype=np.int64)
    for a, b in zip(cu[:-1], cu[1:]):
        n = int(b - a)
        for q0 in range(0, n, q_tile):
            nq = min(q_tile, n - q0)
            items.append((a + q0, nq, 0, 0, a, n, n, 0))
    return np.asarray(items, dtype=np.int32).reshape(-1, 8)


def segment_key_work(cu_seqlens, k_tile=128):
    """Key-centric work items (vr2_attn_bwd_dkv) for NON-causal varlen segments -- the vision tower's windows / per-frame
    sequences when it trains: a key tile of a segment is seen by every query row of that segment.  int32 [n,8]: k_row0, nk,
    q1_row0, q1_len (causal part: none), q2_row0, q2_len (all visible), head0, nheads (1: every head is its own KV head)."""
    items = []
    cu = np.asarray(cu_seqlens, dtype=np.int64)
    for a, b in zip(cu[:-1], cu[1:]):
        n = int(b - a)
        for k0 in range(0, n, k_tile):
            items.append((a + k0, min(k_tile, n - k0), 0, 0, a, n, 0, 1))
    return np.asarray(items, dtype=np.int32).reshape(-1, 8)


def causal_work(n_q, q_row0, prefix_row0, prefix_len, self_row0, q_tile=Q_TILE, q_causal_base=0):
    """n_q consecutive query rows attending [prefix (all visible) | self rows causal]."""
    items = []
    for q0 in range(0, n_q, q_tile):
        nq = min(q_tile, n_q - q0)
        items.append((q_row0 + q0, nq, prefix_row0, prefix_len, self_row0, q_causal_base + n_q, q_causal_base + q0, 0))
    return np.asarray(items, dtype=np.int32).reshape(-1, 8)


def heaviest_first(work):
    """Attention work items (rows of causal_work / segment_work) in descending order of the keys they walk: the grid is
    dispatched in order, so the long items start first and the short ones fill the tail."""
    w = np.asarray(work)
    keys = w[:, 3].astype(np.int64) + np.minimum(w[:, 5].astype(np.int64), w[:, 6].astype(np.int64) + w[:, 1].astype(np.int64))
    return w[np.argsort(-keys, kind="stable")]


def key_work(n_prompt, n_seq, n_comp, k_tile=128, q_split=512, group=1, max_steps=64, lens=None):
    """Key-centric work items (vr2_attn_bwd_dkv) for the unified row space [prompt rows | seq 0 rows | seq 1 rows | ...]:
    a prompt key tile (k_tile <= 128 rows) is seen by the prompt rows from its first row on (causal) and by EVERY completion
    row; a completion key tile by the later rows of its own sequence (causal).  The all-visible range (n_seq * n_comp rows) is
    cut into q_split-row pieces, and an item that would still walk more than max_steps (q head, 64-row query tile) steps is cut
    again by q heads of the GQA group (`group` heads per KV head), so that no workgroup runs much longer than the others; the
    pieces add into the same fp32 dK/dV rows.  Heaviest items first (the grid is dispatched in order).
    lens (optional, one per sequence, <= n_comp): sequence g owns lens[g] rows, packed back to back (completions cut at
    their first EOS: the rows behind it are never computed).
    int32 [n,8]: k_row0, nk, q1_row0, q1_len, q2_row0, q2_len, head0, nheads."""
    assert 1 <= k_tile <= 128
    lens = [n_comp] * n_seq if lens is None else [int(x) for x in lens]
    assert len(lens) == n_seq and all(0 <= x <= n_comp for x in lens)
    items = []

    def add(k0, nk, q1r, q1l, q2r, q2l):
        tiles = (q1l + 63) // 64 + (q2l + 63) // 64
        parts = min(group, max(1, -(-(tiles * group) // max_steps)))
        base, extra = divmod(group, parts)
        h0 = 0
        for i in range(parts):
            nh = base + (1 if i < extra else 0)
            items.append((tiles * nh, (k0, nk, q1r, q1l, q2r, q2l, h0, nh)))
            h0 += nh

    total_c = sum(lens)
    for k0 in range(0, n_prompt, k_tile):
        nk = min(k_tile, n_prompt - k0)
        add(k0, nk, k0, n_prompt - k0, 0, 0)
        for c0 in range(0, total_c, q_split):
            add(k0, nk, 0, 0, n_prompt + c0, min(q_split, total_c - c0))
    base = n_prompt
    for g in range(n_seq):
        for k0 in range(0, lens[g], k_tile):
            nk = min(k_tile, lens[g] - k0)
            add(base + k0, nk, base + k0, lens[g] - k0, 0, 0)
        base += lens[g]
    items.sort(key=lambda it: -it[0])
    return np.asarray([it[1] for it in items], dtype=np.int32).reshape(-1, 8)


@dataclass
class VisionPlan:
    grid_thw: np.ndarray
    n_patches: int
    n_tokens: int
    gather_rows: np.ndarray      # int32 [P]: window-ordered patch row -> source patch row (applied at the input cast)
    window_index: np.ndarray     # int64 [P/unit]
    cu_window: np.ndarray
    cu_full: np.ndarray
    pos_hw: np.ndarray           # int32 [P,2] in WINDOW order
    work_window: np.ndarray      # int32 [n,8]
    work_full: np.ndarray


def make_plan(grid_thw, merge_size=2, window_size=112, patch_size=14) -> VisionPlan:
    g = np.asarray(grid_thw, dtype=np.int64).reshape(-1, 3)
    unit = merge_size * merge_size
    P = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
    wi, cu_w = window_index(g, merge_size, window_size, patch_size)
    cu_f = full_cu_seqlens(g)
    rows = (wi[:, None] * unit + np.arange(unit)[None, :]).reshape(-1)
    pos = position_ids_hw(g, merge_size)[rows]
    return VisionPlan(grid_thw=g, n_patches=P, n_tokens=P // unit, gather_rows=rows.astype(np.int32), window_index=wi,
                      cu_window=cu_w, cu_full=cu_f, pos_hw=pos.astype(np.int32), work_window=segment_work(cu_w),
                      work_full=segment_work(cu_f))
