"""3-D M-RoPE position ids for a [text | video | text] prompt, vectorised numpy (bit-exact integers).

Mirrors what the reference reaches through HF ``get_rope_index``
(transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py:944-1058, called under
/root/reference/train/src/trainer/grpo_trainer.py:725-727 and :783-786): text tokens advance all three axes
by one, a video block uses (t * tokens_per_second * int(second_per_grid_t), h, w) offsets from the running
position and advances it by max(h, w) / merge.
"""
from __future__ import annotations

import numpy as np


def rope_index_1seq(ids, video_token_id, image_token_id, video_grids, second_per_grid_ts, image_grids=(),
                    merge_size=2, tokens_per_second=2):
    """ids: 1-D int array of ONE unpadded sequence -> (pos int64 [3,S], rope_delta int)."""
    ids = np.asarray(ids, dtype=np.int64)
    S = ids.shape[0]
    typ = np.where(ids == video_token_id, 2, np.where(ids == image_token_id, 1, 0))
    # run boundaries
    change = np.flatnonzero(np.diff(typ)) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [S]])
    pos = np.empty((3, S), dtype=np.int64)
    cur = 0
    vi = ii = 0
    for a, b in zip(starts, ends):
        kind = int(typ[a])
        n = int(b - a)
        if kind == 0:
            pos[:, a:b] = cur + np.arange(n, dtype=np.int64)[None, :]
            cur += n
        else:
            if kind == 2:
                t, h, w = (int(x) for x in video_grids[vi])
                sec = second_per_grid_ts[vi] if second_per_grid_ts is not None else 1
                interval = tokens_per_second * int(sec)
                vi += 1
            else:
                t, h, w = (int(x) for x in image_grids[ii])
                interval = 1
                ii += 1
            lh, lw = h // merge_size, w // merge_size
            if t * lh * lw != n:
                raise ValueError(f"Video features and video tokens do not match, tokens: {n}, features: {t * lh * lw}")
            tt, hh, ww = np.meshgrid(np.arange(t, dtype=np.int64) * interval, np.arange(lh, dtype=np.int64),
                                     np.arange(lw, dtype=np.int64), indexing="ij")
            pos[0, a:b] = tt.reshape(-1) + cur
            pos[1, a:b] = hh.reshape(-1) + cur
            pos[2, a:b] = ww.reshape(-1) + cur
            cur += max(h, w) // merge_size
    return pos, int(pos.max()) + 1 - S


def vision_layout(ids, video_token_id, image_token_id, grids, second_per_grid_ts, merge_size=2, tokens_per_second=2):
    """A prompt whose vision items -- video clips AND still images, the `videos=` and `images=` halves of the reference's
    processor call (/root/reference/train/src/trainer/grpo_trainer.py:759-767) -- arrive as ONE list in PROMPT order.

    grids: (t, h, w) per item; second_per_grid_ts: one entry per item, ``None`` marking a still image (HF computes it for videos
    only), or None altogether = every item is a video at 1 s per temporal grid step.
    -> (pos int64 [3,S], rope_delta, vis_pos int64 [vision tokens]: the prompt positions of the items' merged tokens, item after
    item -- the order of a vision plan over ``grids``, so row i of the tower's window-ordered output goes to
    vis_pos[plan.window_index[i]]).
    Raises ValueError, with HF's wording (modeling_qwen2_5_vl.py:1128-1132), when pad-token runs and items do not match."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    grids = [] if grids is None else [tuple(int(x) for x in g) for g in np.asarray(grids).reshape(-1, 3)]
    secs = [1.0] * len(grids) if second_per_grid_ts is None else list(second_per_grid_ts)
    if len(secs) != len(grids):
        raise ValueError(f"{len(grids)} vision items but {len(secs)} second_per_grid_ts entries (None marks a still image)")
    is_img = [s is None for s in secs]
    unit = merge_size * merge_size
    n_img = sum(g[0] * g[1] * g[2] // unit for g, im in zip(grids, is_img) if im)
    n_vid = sum(g[0] * g[1] * g[2] // unit for g, im in zip(grids, is_img) if not im)
    have_img, have_vid = int((ids == image_token_id).sum()), int((ids == video_token_id).sum())
    if have_vid != n_vid:
        raise ValueError(f"Video features and video tokens do not match, tokens: {have_vid}, features: {n_vid}")
    if have_img != n_img and (n_img or any(is_img)):
        raise ValueError(f"Image features and image tokens do not match, tokens: {have_img}, features: {n_img}")
    typ = np.where(ids == video_token_id, 2, np.where((ids == image_token_id) & bool(n_img), 1, 0))
    # the items must come in the order their pad-token runs appear in the prompt
    runs = typ[np.concatenate([[True], typ[1:] != typ[:-1]])] if ids.size else typ
    want = [1 if im else 2 for im in is_img]
    if [int(r) for r in runs if r] != want:
        raise ValueError("vision items are not in prompt order: the prompt has runs "
                         f"{['image' if r == 1 else 'video' for r in runs if r]}, the items are {['image' if x == 1 else 'video' for x in want]}")
    pos, delta = rope_index_1seq(ids, video_token_id, image_token_id if n_img else -1, [g for g, im in zip(grids, is_img) if not im],
                                 [s for s in secs if s is not None], image_grids=[g for g, im in zip(grids, is_img) if im],
                                 merge_size=merge_size, tokens_per_second=tokens_per_second)
    return pos, delta, np.flatnonzero(typ != 0)
