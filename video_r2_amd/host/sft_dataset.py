"""Stage-1 supervised rows -> token ids + labels (the data module of the SFT entry point).

What `SupervisedDataset.__getitem__` hands the reference's SFT trainer (/root/reference/train/src/dataset/sft_dataset.py:58-206):
a LLaVA-style row {"video" | "image": file(s), "conversations": [{"from": "human", "value": ...}, {"from": "gpt", ...}, ...]}
becomes ONE token sequence laid out piece by piece --

    <|im_start|>system\\nYou are a helpful assistant.<|im_end|>\\n                           labels: ignored        (:127-133)
    per (user, answer) pair:
      <|im_start|>{role}\\n{content}<|im_end|>\\n<|im_start|>{answer role}\\n            labels: ignored        (:139,167-176)
      {answer}<|im_end|>\\n                                                               labels: the ids themselves (:140,165)

-- each piece tokenised on its own (`add_special_tokens=False`), the `<video>` / `<image>` tag of a turn (with the newline on
either side of it) replaced by `<|vision_start|><|video_pad|><|vision_end|>` (data_utils.py:16-25) and every pad placeholder
expanded to one id per merged patch of its clip / picture (the HF processor's expansion).  Roles: "human" -> "user", anything
else passes through unchanged (data_utils.py:27-38: the map names "human" and "assistant" only).  Media files resolve against
--image_folder, subtitles by the video's relative path with an .srt suffix under --video_subtitles_folder (:96-103); the pixel
and fps knobs of the data arguments travel in every media element (data_utils.py:73-120).

The reference hands ALL of a row's media to the processor together with the first turn that names a placeholder; here the
placeholders consume the row's media in order wherever they stand, and a count mismatch is an error (the HF processor's).
"""
from __future__ import annotations

import json
import os
import re

import numpy as np
import torch

from . import frontend as fe
from . import processor as hp
from .dataset import SYSTEM_MESSAGE, _media_element

IGNORE_INDEX = -100                                        # /root/reference/train/src/constants.py:1


def media_tag_to_placeholder(text, is_video):
    tag, pad = ("<video>", fe.VIDEO_PAD) if is_video else ("<image>", fe.IMAGE_PAD)
    return re.sub(r"\n?" + re.escape(tag) + r"\n?", fe.VISION_START + pad + fe.VISION_END, text)


def to_turns(conversations, is_video):
    return [{"role": {"human": "user"}.get(t["from"], t["from"]), "content": media_tag_to_placeholder(t["value"], is_video)}
            for t in conversations]


class SFTDataset:
    """len() / [i] over the rows of a JSON list file (or an in-memory list) -> {"turns": [...], "media": [...], "is_video": bool}.
    `data_args` carries the attributes the flags of train_sft.sh set (image_folder, video_subtitles_folder, video_max_pixels, fps ...)."""

    def __init__(self, data_path, data_args):
        if isinstance(data_path, (str, os.PathLike)):
            with open(data_path, "r") as f:
                data_path = json.load(f)
        self.rows = data_path
        g = lambda name, default=None: getattr(data_args, name, default)      # noqa: E731
        self.image = (g("image_min_pixels", 3136), g("image_max_pixels", 12845056), g("image_resized_width"), g("image_resized_height"))
        self.video = (g("video_min_pixels", 100352), g("video_max_pixels", 602112), g("video_resized_width"), g("video_resized_height"))
        self.folder = g("image_folder") or ""
        self.subtitles = g("video_subtitles_folder")
        self.fps = g("fps", 1.0)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        row = self.rows[i]
        media, is_video = [], False
        if "image" in row:
            for f in (row["image"] if isinstance(row["image"], list) else [row["image"]]):
                if not os.path.exists(f) and not f.startswith("http"):
                    f = os.path.join(self.folder, f)
                media.append(_media_element("image", f, *self.image))
        elif "video" in row:
            is_video = True
            for f in (row["video"] if isinstance(row["video"], list) else [row["video"]]):
                srt = None
                if self.subtitles:
                    cand = os.path.splitext(os.path.join(self.subtitles, f))[0] + ".srt"
                    srt = cand if os.path.exists(cand) else None
                path = f if f.startswith("http") else os.path.join(self.folder, f)
                mn, mx, w, h = self.video
                media.append(_media_element("video", path, mn, mx, w, h, subtitles=srt, fps=self.fps))
        return {"turns": to_turns(row["conversations"], is_video), "media": media, "is_video": is_video}


class SFTFrontEnd:
    """processing_fn(example) for QwenSFTTrainer: dataset example -> input_ids, labels, uint8 clips (on pinned memory)."""

    def __init__(self, cfg, tokenizer, clip_opener=fe.open_clip, model_dir=None):
        self.cfg, self.tok, self.open = cfg, tokenizer, clip_opener
        self.image_pixels = fe.processor_pixel_range(model_dir)
        self._pads = re.compile("(" + re.escape(fe.VIDEO_PAD) + "|" + re.escape(fe.IMAGE_PAD) + ")")

    def _tokens_of(self, n_frames, h, w):
        v = self.cfg.vision
        t = (n_frames + v.temporal_patch_size - 1) // v.temporal_patch_size
        return t * (h // v.patch_size) * (w // v.patch_size) // v.spatial_merge_size ** 2

    def __call__(self, example):
        videos, secs, counts, pads = [], [], [], []
        for c in example["media"]:
            if c["type"] == "image":
                frame = fe.read_image(c, self.image_pixels)
                videos.append((frame.pin_memory() if torch.cuda.is_available() else frame, None))
                secs.append(None)
                counts.append(self._tokens_of(1, int(frame.shape[2]), int(frame.shape[3])))
                pads.append((fe.IMAGE_PAD, self.cfg.image_token_id))
            else:
                frames, sfps, dims = fe.read_video(c, self.open(c["video"]))
                videos.append((frames.pin_memory() if torch.cuda.is_available() else frames, dims))
                secs.append(hp.second_per_grid_t(self.cfg.vision.temporal_patch_size, sfps))
                counts.append(self._tokens_of(frames.shape[0], *dims))
                pads.append((fe.VIDEO_PAD, self.cfg.video_token_id))
        ids, labels, used = [], [], 0

        def piece(text, supervised):
            nonlocal used
            parts = self._pads.split(text)
            out = []
            for j, part in enumerate(parts):
                if j % 2 == 0:
                    out.extend(self.tok.encode(part))
                    continue
                if used >= len(pads) or pads[used][0] != part:
                    raise ValueError(f"placeholder {part} number {used + 1} of the conversation has no matching media element "
                                     f"({len(pads)} media element(s): {[p[0] for p in pads]})")
                out.extend([pads[used][1]] * counts[used])          # one pad id per merged patch (processing_qwen2_5_vl.py:138-154)
                used += 1
            ids.extend(out)
            labels.extend(out if supervised else [IGNORE_INDEX] * len(out))

        if SYSTEM_MESSAGE:
            piece(f"{fe.IM_START}system\n{SYSTEM_MESSAGE}{fe.IM_END}\n", False)
        turns = example["turns"]
        for j in range(0, len(turns) - 1, 2):
            user, answer = turns[j], turns[j + 1]
            piece(f"{fe.IM_START}{user['role']}\n{user['content']}{fe.IM_END}\n{fe.IM_START}{answer['role']}\n", False)
            piece(f"{answer['content']}{fe.IM_END}\n", True)
        if used != len(pads):
            raise ValueError(f"{len(pads)} media element(s) but {used} placeholder(s) in the conversation")
        return {"input_ids": np.asarray(ids, dtype=np.int64), "labels": np.asarray(labels, dtype=np.int64), "videos": videos,
                "second_per_grid_ts": secs or None}


def make_supervised_data_module(model_id, processor, data_args):
    return {"train_dataset": SFTDataset(data_args.data_path, data_args), "eval_dataset": None}
