"""Evaluation-harness side of the engine: ``generate_until`` with the request contract of the reference's
lmms-eval model class (/root/reference/eval/lmms-eval/lmms_eval/models/qwen2_5_vl.py:181-357), over ``VideoR2Model``'s kernels.

What the harness hands over per request (``Instance.args``): (context, gen_kwargs, doc_to_visual, doc_id, task, split); the
visuals of a document are ``doc_to_visual(task_dict[task][split][doc_id])``.  Per request this class builds the chat
[system prompt, user: the document's videos + the context text], samples frames in two stages as the reference does -- the
qwen-vl frame policy (fps / min / max frames, pixel budget: host/frames.py), then at most ``max_num_frames`` of those by
``np.linspace(0, n - 1, max_num_frames, dtype=int)`` plus the last frame when it is not among them (:285-291) -- encodes the
uint8 frames on the device (resize, normalise, patchify: Engine.encode_videos), prefills, decodes greedily when the
temperature is 0 (the harness default, :305-320) or with temperature + top-k otherwise, cuts the text at the first ``until``
string ("\\n\\n" is dropped from the list as in :212-213) and returns the answers in request order.

When ``lmms_eval`` is importable the class is registered as the model ``video_r2_amd``; without it (this container) it is a
plain class with the same methods, which is what tests/test_eval_adapter_gpu.py drives with stand-in requests.
"""
from __future__ import annotations

import re

import numpy as np
import torch

from .host import frames as hf
from .host import processor as hp
from .host.frontend import (IMAGE_PAD, VIDEO_PAD, ProcessingFrontEnd, frame_range, load_tokenizer, open_clip, read_image, render_chat)

VIDEO_SUFFIXES = (".mp4", ".avi", ".mov", ".mkv", ".webm", ".npy", ".npz", ".pt")


def eval_frame_plan(ele, total_frames, video_fps, height, width, max_num_frames):
    """-> (source frame indices in the order they are fed, sample_fps of stage one, (H', W') to resize to).
    Stage one = the training front end's policy (vision_process.py:425-462,494-495,780-801); stage two = the harness's own
    cut to max_num_frames (qwen2_5_vl.py:285-291): positions by truncating linspace, the last frame appended if missing, so
    a clip shorter than max_num_frames repeats frames exactly as the reference does."""
    first, last, total = frame_range(ele, total_frames, video_fps)
    n1 = hf.smart_nframes(ele, total_frames=total, video_fps=video_fps)
    idx1 = np.asarray(hf.frame_indices(total, n1, start=first, end=last).tolist(), dtype=np.int64)
    pos = np.linspace(0, n1 - 1, max_num_frames, dtype=int)
    if n1 - 1 not in pos:
        pos = np.append(pos, n1 - 1)
    dims = hf.video_resize_dims(n1, height, width, ele)
    return idx1[pos], hf.sample_fps(n1, total, video_fps), dims


class VideoR2EvalModel:
    def __init__(self, pretrained=None, device="cuda:0", batch_size=1, min_pixels=256 * 28 * 28, max_pixels=1605632,
                 max_num_frames=32, fps=None, system_prompt="You are a helpful assistant.", reasoning_prompt=None,
                 model=None, tokenizer=None, clip_opener=open_clip, task_dict=None, **kwargs):
        if kwargs:
            raise ValueError(f"Unexpected kwargs: {kwargs}")
        if model is None:
            from .model import VideoR2Model
            model = VideoR2Model.from_pretrained(pretrained, device=device)
        self._model = model
        self._tokenizer = tokenizer if tokenizer is not None else load_tokenizer(pretrained, model.cfg)
        self.front = ProcessingFrontEnd(model.cfg, self._tokenizer, clip_opener, model_dir=pretrained if isinstance(pretrained, str) else None)
        self.open = clip_opener
        self.min_pixels, self.max_pixels, self.max_num_frames, self.fps = int(min_pixels), int(max_pixels), int(max_num_frames), fps
        self.system_prompt = system_prompt
        self.reasoning_prompt = reasoning_prompt.replace("\\n", "\n") if reasoning_prompt else None
        self.batch_size_per_gpu = int(batch_size)
        self.task_dict = task_dict if task_dict is not None else {}
        self._rank, self._world_size = 0, 1

    # -- the properties the harness reads
    @property
    def model(self):
        return self._model

    @property
    def tokenizer(self):
        return self._tokenizer

    @property
    def config(self):
        return self._model.cfg

    @property
    def eot_token_id(self):
        return self._model.cfg.eos_token_id

    @property
    def batch_size(self):
        return self.batch_size_per_gpu

    @property
    def device(self):
        return self._model.device

    @property
    def rank(self):
        return self._rank

    @property
    def world_size(self):
        return self._world_size

    def loglikelihood(self, requests):
        raise NotImplementedError("Loglikelihood is not implemented for the Video-R2 engine adapter (as in the reference class)")

    def generate_until_multi_round(self, requests):
        raise NotImplementedError("multi-round generation is not implemented (as in the reference class)")

    # -- one request
    def _video_element(self, path):
        ele = {"type": "video", "video": path, "max_pixels": self.max_pixels, "min_pixels": self.min_pixels}
        if self.fps is not None:
            ele["fps"] = self.fps
        return ele

    def _clip(self, ele):
        clip = self.open(ele["video"])
        probe = np.asarray(clip.get([0]))                                     # [1,H,W,3]: the frame size
        idx, sfps, dims = eval_frame_plan(ele, len(clip), float(clip.fps), probe.shape[1], probe.shape[2], self.max_num_frames)
        fr = np.asarray(clip.get(idx.tolist()), dtype=np.uint8)              # [n,H,W,3]
        frames = torch.from_numpy(fr).permute(0, 3, 1, 2).contiguous()
        return frames, sfps, dims

    def _prepare(self, context, visuals):
        """one request -> (prompt ids, clips for Engine.encode_videos, second_per_grid_ts, the context as sent)"""
        cfg = self._model.cfg
        context = context.replace("<image>", "")
        if self.reasoning_prompt:
            context = context.strip() + self.reasoning_prompt
        content, clips, secs, counts, pads = [], [], [], [], []
        for v in visuals:
            if isinstance(v, str) and v.lower().endswith(VIDEO_SUFFIXES):
                ele = self._video_element(v)
                frames, sfps, dims = self._clip(ele)
                content.append(ele)
                clips.append(dict(frames_u8=frames, resize_to=dims))
                secs.append(hp.second_per_grid_t(cfg.vision.temporal_patch_size, sfps))
                counts.append(self.front.video_tokens(frames.shape[0], *dims))
                pads.append(cfg.video_token_id)
            elif hasattr(v, "convert") and hasattr(v, "save"):
                # a PIL image: the harness sends it as an RGB JPEG data URI (qwen2_5_vl.py:239-245), so the model sees the JPEG
                # round trip's pixels, resized by fetch_image under the adapter's pixel range
                import base64
                import io
                buf = io.BytesIO()
                v.convert("RGB").save(buf, format="JPEG")
                ele = {"type": "image", "image": "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode("utf-8"),
                       "max_pixels": self.max_pixels, "min_pixels": self.min_pixels}
                frame = read_image(ele, self.front.image_pixels)
                content.append(ele)
                clips.append(dict(frames_u8=frame, resize_to=None))
                secs.append(None)                                        # None = still image
                counts.append(self.front.video_tokens(1, int(frame.shape[2]), int(frame.shape[3])))
                pads.append(cfg.image_token_id)
            else:     # (the harness class drops such a visual without a word and answers from the text alone: refused here instead)
                raise NotImplementedError(f"visual of type {type(v).__name__}: the adapter takes video files ({', '.join(VIDEO_SUFFIXES)}) "
                                          "and PIL images, as the reference's model class does")
        messages = [{"role": "system", "content": self.system_prompt},
                    {"role": "user", "content": content + [{"type": "text", "text": context}]}]
        pieces = re.split("(" + re.escape(VIDEO_PAD) + "|" + re.escape(IMAGE_PAD) + ")", render_chat(messages, add_generation_prompt=True))
        ids, k = [], 0
        for j, piece in enumerate(pieces):
            if j % 2 == 0:
                ids.extend(self._tokenizer.encode(piece))
            else:
                ids.extend([pads[k]] * counts[k])
                k += 1
        return np.asarray(ids, dtype=np.int64), clips, secs or None, context

    def _answers(self, prepared, gen_kwargs):
        """the requests of one batch (same generation arguments): ONE pass of the vision tower over all their clips, ONE prompt
        prefill, and their sequences decoded together (the decode is weight-streaming bound: b rows cost little more than one)"""
        cfg, eng = self._model.cfg, self._model.engine
        kw = {"max_new_tokens": 128, "temperature": 0.0, "top_p": None, "num_beams": 1, **gen_kwargs}
        if (kw.get("num_beams") or 1) != 1:
            raise ValueError(f"num_beams={kw['num_beams']}: the engine samples or decodes greedily, it has no beam search")
        sample = float(kw["temperature"] or 0.0) > 0.0
        idss = [p[0] for p in prepared]
        clips = [p[1] for p in prepared]
        enc = eng.encode_videos([c for cl in clips for c in cl], per_prompt=[len(cl) for cl in clips])
        dec = dict(temperature=float(kw["temperature"]) if sample else 1.0, top_k=int(kw.get("top_k") or 50) if sample else 1,
                   seed=int(kw.get("seed", 0)), eos_token_id=cfg.eos_token_id,
                   top_p=(kw.get("top_p") if sample else None), min_p=(kw.get("min_p") if sample else None),
                   repetition_penalty=kw.get("repetition_penalty") or 1.0)
        if len(prepared) == 1:
            emb, plan = enc[0]
            st = eng.prefill(self._model.text, idss[0], None if emb is None else emb.clone(), plan, prepared[0][2], tag="pol")
            toks, _ = eng.decode(self._model.text, st, 1, int(kw["max_new_tokens"]), **dec)
        else:
            slots = eng.prefix_slots(len(prepared), max(len(i) for i in idss), tag="pol")
            states = eng.prefill_batch(self._model.text, [(ids, emb, plan, p[2]) for ids, (emb, plan), p in zip(idss, enc, prepared)],
                                       slots, tag="pol")
            toks, _ = eng.decode(self._model.text, states, 1, int(kw["max_new_tokens"]), **dec)
        eng.sync()
        out = []
        self.last_token_rows = []            # the batch's answers as token ids (up to the stop id): what the texts were decoded from
        for row in toks.cpu().numpy():
            stop = np.flatnonzero(row == cfg.eos_token_id)
            row = row[:stop[0]] if stop.size else row
            self.last_token_rows.append(row.copy())
            out.append(self._tokenizer.decode(row.tolist(), skip_special_tokens=True))
        return out

    def generate_until(self, requests):
        """-> list[str], one answer per request, in request order.  Consecutive requests with the same generation arguments are
        answered batch_size at a time."""
        out = [None] * len(requests)
        i = 0
        while i < len(requests):
            gen_kwargs = requests[i].args[1]
            j = i
            while j < len(requests) and j - i < self.batch_size_per_gpu and requests[j].args[1] == gen_kwargs:
                j += 1
            until = gen_kwargs.get("until", [self._tokenizer.decode([self.eot_token_id], skip_special_tokens=False)])
            if isinstance(until, str):
                until = [until]
            elif not isinstance(until, list):
                raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str, list], but got {type(until)}")
            until = [u for u in until if u != "\n\n"]
            kw = {k: v for k, v in gen_kwargs.items() if k != "until"}
            prepared = []
            for req in requests[i:j]:
                context, _, doc_to_visual, doc_id, task, split = req.args
                prepared.append(self._prepare(context, list(doc_to_visual(self.task_dict[task][split][doc_id]))))
            for k, ans in enumerate(self._answers(prepared, kw)):
                for term in until:
                    if len(term) > 0:
                        ans = ans.split(term)[0]
                out[i + k] = ans
                hook = getattr(self, "cache_hook", None)
                if hook is not None:
                    hook.add_partial("generate_until", (prepared[k][3], gen_kwargs), ans)
            i = j
        return out


try:                                                      # the harness is not installed here; register when it is
    from lmms_eval.api.model import lmms as _lmms
    from lmms_eval.api.registry import register_model as _register

    @_register("video_r2_amd")
    class VideoR2LmmsEval(VideoR2EvalModel, _lmms):
        def __init__(self, *a, **k):
            _lmms.__init__(self)
            VideoR2EvalModel.__init__(self, *a, **k)
except ImportError:
    VideoR2LmmsEval = None
