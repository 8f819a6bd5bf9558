"""GRPO training rows -> chat-format prompts (the data module of the entry point).

What `make_grpo_data_module` / `GRPODataset.__getitem__` give the reference trainer
(/root/reference/train/src/dataset/grpo_dataset.py:72-195): one JSON row becomes

    {"prompt": [ {"role": "system", "content": SYSTEM_MESSAGE},
                 {"role": "user",   "content": [ <one element per image / video>, {"type": "text", "text": question} ]} ],
     "assistant": {"role": "assistant", "content": "<think>...</think><answer>..</answer>"},
     "temporal_captions": {...}, "temporal_grounding": {...}}

with the `<video>` / `<image>` placeholder (and the newline next to it) removed from the question (:15-18), media paths
resolved against --image_folder (:119-152), the subtitle file looked up by the video's relative path with an .srt suffix
under --video_subtitles_folder (:142-148) and the pixel / fps knobs of the data arguments copied into every media element
(:36-69).  Every key of the returned dict other than "prompt" is forwarded to the reward functions as a column
(grpo_trainer.py:867-868), so the names are part of the plugin contract.
"""
from __future__ import annotations

import copy
import json
import os
import re

SYSTEM_MESSAGE = "You are a helpful assistant."            # /root/reference/train/src/constants.py:12
_ROLES = {"human": "user", "gpt": "assistant"}             # any other "from" value passes through unchanged (:21, :27)


def strip_media_tag(text, is_video):
    tag = "<video>" if is_video else "<image>"
    return re.sub(r"\n?" + re.escape(tag) + r"\n?", "", text)


def to_chat_turns(conversations, is_video):
    # NB the reference's role map only names "human" (and the identity "assistant"); "gpt" passes through as the role string.
    # Nothing downstream reads the role of the answer turn (rewards read ["content"]), the prompt only uses the first turn.
    return [{"role": {"human": "user"}.get(t["from"], t["from"]), "content": strip_media_tag(t["value"], is_video)}
            for t in conversations]


def _media_element(kind, path, min_pixels, max_pixels, width, height, **extra):
    ele = {"type": kind, kind: path, **extra, "min_pixels": min_pixels, "max_pixels": max_pixels}
    if width is not None and height is not None:
        ele["resized_width"], ele["resized_height"] = width, height
    return ele


class GRPODataset:
    """len() / [i] over the rows of a JSON list file (or an in-memory list).  `data_args` needs the attributes the flags of
    train_grpo.sh set: image_folder, video_subtitles_folder, video_min_pixels, video_max_pixels, fps, and optionally
    image_min_pixels / image_max_pixels / {image,video}_resized_{width,height}."""

    def __init__(self, data_path, data_args, model_id=None, processor=None):
        if isinstance(data_path, (str, os.PathLike)):
            with open(data_path, "r") as f:
                rows = json.load(f)
        else:
            rows = data_path
        self.rows = rows
        self.data_args = data_args
        self.model_id = model_id
        self.processor = processor
        g = lambda name, default=None: getattr(data_args, name, default)      # noqa: E731
        self.image = (g("image_min_pixels", 3136), g("image_max_pixels", 12845056), g("image_resized_width"), g("image_resized_height"))
        self.video = (g("video_min_pixels", 100352), g("video_max_pixels", 602112), g("video_resized_width"), g("video_resized_height"))
        self.folder = g("image_folder") or ""
        self.subtitles = g("video_subtitles_folder")
        self.fps = g("fps", 1.0)

    def __len__(self):
        return len(self.rows)

    def _resolve(self, name, keep_existing):
        if name.startswith("http") or (keep_existing and os.path.exists(name)):
            return name
        return os.path.join(self.folder, name)

    def __getitem__(self, i):
        row = self.rows[i]
        media, is_video = [], False
        if "image" in row:
            files = row["image"] if isinstance(row["image"], list) else [row["image"]]
            for f in files:
                media.append(_media_element("image", self._resolve(f, keep_existing=True), *self.image))
        elif "video" in row:
            is_video = True
            files = row["video"] if isinstance(row["video"], list) else [row["video"]]
            keys = row.get("key_frame", "")
            keys = keys if isinstance(keys, list) else [keys]
            for f, key in zip(files, keys):
                srt = None
                if self.subtitles:
                    cand = os.path.splitext(os.path.join(self.subtitles, f))[0] + ".srt"
                    srt = cand if os.path.exists(cand) else None
                mn, mx, w, h = self.video
                media.append(_media_element("video", self._resolve(f, keep_existing=False), mn, mx, w, h,
                                            subtitles=srt, key_frame=key, fps=self.fps))
        turns = copy.deepcopy(to_chat_turns(row["conversations"], is_video))
        prompt = [{"role": "user", "content": media + [{"type": "text", "text": turns[0]["content"]}]}]
        if SYSTEM_MESSAGE:
            prompt.insert(0, {"role": "system", "content": SYSTEM_MESSAGE})
        return {"prompt": prompt, "assistant": turns[1], "temporal_captions": row.get("temporal_captions", {}),
                "temporal_grounding": row.get("temporal_grounding", {})}


def make_grpo_data_module(model_id, processor, data_args):
    return {"train_dataset": GRPODataset(data_args.data_path, data_args, model_id=model_id, processor=processor),
            "eval_dataset": None}
