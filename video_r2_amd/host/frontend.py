"""Host front end of the rollout: dataset example -> what the engine needs to roll a prompt group out.

The reference does this inside `_generate_and_score_completions` (/root/reference/train/src/trainer/grpo_trainer.py:755-767):
chat template -> `process_vision_info` (read video, pick frames, burn timestamps / subtitles in, resize) -> HF processor
(video-pad expansion, tokenise, normalise + patchify).  Here the same steps produce

    {"input_ids": int64 [S] (unpadded prompt), "videos": [(uint8 frames [n,3,H,W], (H', W') to resize to), ...],
     "second_per_grid_ts": [float, ...], "decode": callable(completion ids [G,C]) -> list[str]}

("videos" = the prompt's vision items in prompt order: a still image -- dataset rows with an "image" key, grpo_dataset.py:113-125,
the `images=` half of the processor call -- is an item of ONE frame, already resized by PIL as `fetch_image` does it, with None
as its second_per_grid_ts entry and <|image_pad|> tokens in the prompt)

and the uint8 frames go to the device as they are: the antialiased bicubic resize, rescale + normalise and patchify run there
(Engine.resize_frames / encode_frames, bit-identical to torchvision + the HF video processor).  Host-side pieces, each after
the reference lines it follows:

  * frame range / count / indices / sample fps              vision_process.py:549-605, :425-462, :632, :664   (host/frames.py)
  * HH:MM:SS stamp top-left, subtitle box bottom-centre      vision_process.py:67-133, :137-309 (PIL; on when the env says so)
  * resize dimensions under the per-frame pixel budget       vision_process.py:780-801                         (host/frames.py)
  * Qwen2.5-VL chat layout + one video-pad token per merged patch   HF processing_qwen2_5_vl.py:138-154; TRL maybe_apply_chat_template
  * second_per_grid_ts = temporal_patch_size / sample_fps    HF processing_qwen2_5_vl.py

Video files: decoded through whichever of torchcodec / decord / OpenCV is importable (the reference's backend order,
:748-768); none of them exists in the offline build container, so tests use array clips (.npy / .npz / .pt holding uint8
[T,H,W,3] frames and an fps).  Tokeniser: the model directory's tokenizer.json when there is one, else a byte-level
stand-in (tests, random-init dry runs).  PIL text rendering has no runnable reference here ("parity unpinned" for pixels
under the overlays; the integer helpers -- time formatting, SRT parsing, cue lookup, colour choice -- are tested as such).
"""
from __future__ import annotations

import json
import math
import os
import re

import numpy as np
import torch

from . import frames as hf
from . import processor as hp

IM_START, IM_END = "<|im_start|>", "<|im_end|>"
VISION_START, VISION_END = "<|vision_start|>", "<|vision_end|>"
VIDEO_PAD, IMAGE_PAD, END_OF_TEXT = "<|video_pad|>", "<|image_pad|>", "<|endoftext|>"


def _env_flag(name):
    return os.getenv(name, "False").strip().lower() == "true"


# ---------------------------------------------------------------------------------------------- clip sources
class ArrayClip:
    """uint8 frames held in a file: .npy ([T,H,W,3]; fps from a '<file>.json' {"fps": f} beside it, else 30), .npz
    (arrays 'frames' and 'fps') or .pt ({'frames': uint8 tensor, 'fps': float}, loaded with weights_only=True)."""

    def __init__(self, path):
        ext = os.path.splitext(path)[1].lower()
        fps = 30.0
        if ext == ".npy":
            fr = np.load(path, mmap_mode="r")
            side = path + ".json"
            if os.path.exists(side):
                with open(side) as f:
                    fps = float(json.load(f)["fps"])
        elif ext == ".npz":
            z = np.load(path)
            fr, fps = z["frames"], float(z["fps"]) if "fps" in z else fps
        else:
            d = torch.load(path, map_location="cpu", weights_only=True)
            fr, fps = d["frames"].numpy(), float(d.get("fps", fps))
        if fr.ndim != 4 or fr.dtype != np.uint8:
            raise ValueError(f"{path}: want uint8 [T,H,W,3] frames, got {fr.dtype} {fr.shape}")
        if fr.shape[-1] != 3 and fr.shape[1] == 3:
            fr = np.transpose(fr, (0, 2, 3, 1))
        self.frames, self.fps = fr, fps

    def __len__(self):
        return int(self.frames.shape[0])

    def get(self, idx):
        return np.ascontiguousarray(self.frames[np.asarray(idx)])


class DecordClip:
    def __init__(self, path):
        import decord
        self.vr = decord.VideoReader(path)
        self.fps = float(self.vr.get_avg_fps())

    def __len__(self):
        return len(self.vr)

    def get(self, idx):
        return self.vr.get_batch(list(idx)).asnumpy()


class TorchcodecClip:
    def __init__(self, path):
        from torchcodec.decoders import VideoDecoder
        self.dec = VideoDecoder(path, num_ffmpeg_threads=int(os.environ.get("TORCHCODEC_NUM_THREADS", 8)), dimension_order="NHWC")
        self.fps = float(self.dec.metadata.average_fps)

    def __len__(self):
        return int(self.dec.metadata.num_frames)

    def get(self, idx):
        return self.dec.get_frames_at(indices=list(idx)).data.cpu().numpy()


class OpenCVClip:
    def __init__(self, path):
        import cv2
        self.cv2, self.cap = cv2, cv2.VideoCapture(path)
        if not self.cap.isOpened():
            raise IOError(f"cannot open {path}")
        self.fps = float(self.cap.get(cv2.CAP_PROP_FPS))

    def __len__(self):
        return int(self.cap.get(self.cv2.CAP_PROP_FRAME_COUNT))

    def get(self, idx):
        out = []
        for i in idx:
            self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, int(i))
            ok, bgr = self.cap.read()
            if not ok:
                raise IOError(f"frame {i} unreadable")
            out.append(bgr[:, :, ::-1])
        return np.stack(out)


def open_clip(path):
    """Array clips by extension; real containers through the first importable decoder in the reference's order."""
    if path.startswith("file://"):
        path = path[7:]
    if os.path.splitext(path)[1].lower() in (".npy", ".npz", ".pt"):
        return ArrayClip(path)
    forced = os.getenv("FORCE_QWENVL_VIDEO_READER")
    order = [forced] if forced else ["torchcodec", "decord", "opencv"]
    errors = []
    for name in order:
        try:
            return {"torchcodec": TorchcodecClip, "decord": DecordClip, "opencv": OpenCVClip}[name](path)
        except Exception as e:  # ImportError when the package is absent, decoder errors otherwise: try the next one
            errors.append(f"{name}: {type(e).__name__}: {e}")
    raise RuntimeError(f"no video decoder could read {path} ({'; '.join(errors)}); install torchcodec / decord / opencv or "
                       "convert the clip to an array file (.npy / .npz / .pt)")


# ---------------------------------------------------------------------------------------------- frame selection
def frame_range(ele, total_frames, video_fps):
    """(first, last, count) of the frames between the optional 'video_start' / 'video_end' seconds (vision_process.py:549-605)."""
    if video_fps <= 0:
        raise ValueError("video_fps must be a positive number")
    if total_frames <= 0:
        raise ValueError("total_frames must be a positive integer")
    t0, t1 = ele.get("video_start"), ele.get("video_end")
    if t0 is None and t1 is None:
        return 0, total_frames - 1, total_frames
    dur = total_frames / video_fps
    first = math.ceil(max(0.0, min(t0, dur)) * video_fps) if t0 is not None else 0
    last = min(math.floor(max(0.0, min(t1, dur)) * video_fps), total_frames - 1) if t1 is not None else total_frames - 1
    if first >= last:
        raise ValueError(f"Invalid time range: start frame {first} is not before end frame {last} "
                         f"({total_frames} frames @ {video_fps} fps)")
    return first, last, last - first + 1


# ---------------------------------------------------------------------------------------------- overlays (PIL)
def format_hhmmss(seconds):
    s = int(seconds)
    return f"{s // 3600:02d}:{s % 3600 // 60:02d}:{s % 60:02d}"


def stamp_colour(frame):
    """White on a dark or red-ish top-left patch (40 x 200 px), red otherwise (vision_process.py:77-99)."""
    patch = frame[:min(40, frame.shape[0]), :min(200, frame.shape[1])].astype(np.float32)
    r, g, b = (float(patch[..., c].mean()) for c in range(3))
    dark = 0.2126 * r + 0.7152 * g + 0.0722 * b < 80.0
    reddish = r > 120.0 and r - max(g, b) > 40.0
    return (255, 255, 255) if (dark or reddish) else (255, 0, 0)


def _font(px):
    from PIL import ImageFont
    try:
        return ImageFont.truetype("DejaVuSans.ttf", px)
    except Exception:
        return ImageFont.load_default()


def draw_timestamp(frame, text):
    """HH:MM:SS in the top-left corner; font 5 % of the short side clamped to [16, 72] px, margin font/4 (>= 5), black outline
    font/12 (>= 1) (vision_process.py:102-133)."""
    from PIL import Image, ImageDraw
    img = Image.fromarray(np.ascontiguousarray(frame), mode="RGB")
    px = max(16, min(72, int(round(min(frame.shape[:2]) * 0.05))))
    m = max(5, px // 4)
    ImageDraw.Draw(img).text((m, m), text, fill=stamp_colour(frame), font=_font(px), stroke_width=max(1, px // 12), stroke_fill=(0, 0, 0))
    return np.asarray(img)


_SRT_CLOCK = re.compile(r"^\s*([+-]?\d+)\s*:\s*([+-]?\d+)\s*:\s*([+-]?\d+)\s*,\s*([+-]?\d+)\s*$")
_SRT_TAG = re.compile(r"</?[biu]>", re.IGNORECASE)


def _srt_clock(field):
    """'HH:MM:SS,mmm' -> seconds (the milliseconds field is an integer count of ms whatever its width); None if malformed."""
    m = _SRT_CLOCK.match(field)
    if m is None:
        return None
    hh, mm, ss, ms = (int(g) for g in m.groups())
    return hh * 3600 + mm * 60 + ss + ms / 1000.0


def parse_srt(path):
    """SubRip file -> [(start s, end s, [text lines])] ordered by start time.

    Behaviour of the reference's parser (vision_process.py:145-186): blocks are separated by blank lines; the timing line is the
    first of a block's first three lines that holds '-->' (an index line may or may not precede it); a block with no such line,
    with a timing line that is not exactly two 'HH:MM:SS,mmm' clocks, or with no text is dropped; <i> <b> <u> tags are removed;
    a UTF-8 byte-order mark and CR / CRLF line ends are accepted."""
    with open(path, "rb") as f:
        data = f.read()
    raw = None
    for codec in ("utf-8", "utf-8-sig"):                                # the reference's order: plain first, BOM-aware second
        try:
            raw = data.decode(codec)
            break
        except UnicodeDecodeError as err:
            last = err
    if raw is None:
        raise last
    cues = []
    for block in re.split(r"\n{2,}", re.sub(r"\r\n?", "\n", raw)):
        rows = block.strip().split("\n")
        timing = [k for k, row in enumerate(rows[:3]) if "-->" in row]
        if not timing:
            continue
        clocks = [_srt_clock(f) for f in rows[timing[0]].split("-->")]
        if len(clocks) != 2 or None in clocks:
            continue
        text = [_SRT_TAG.sub("", row.strip()) for row in rows[timing[0] + 1:] if row.strip()]
        if text:
            cues.append((clocks[0], clocks[1], text))
    return sorted(cues, key=lambda c: c[0])                             # stable: equal starts keep file order


class SubtitleTrack:
    """Which cue is on screen at a frame time.  Stateless: cue start / end times live in two arrays and a query is a
    ``numpy.searchsorted`` over the starts, so a whole clip's frame times are resolved in one call (``texts``).

    Rule (what vision_process.py:189-212 computes for the non-decreasing frame times the reader feeds it, tol = 50 ms): the
    candidate is the LAST cue whose start is <= t + tol; it is shown while t < its end + tol.  Overlapping cues: the later
    start wins and an earlier cue that is still running behind an ended later one is not shown.  A candidate whose start lies
    ahead of t although it has already "ended" (end < start - tol: a malformed cue) hands over to the cue before it."""

    def __init__(self, cues, tol=0.05):
        self.text = ["\n".join(c[2]) for c in cues]
        self.start = np.asarray([c[0] for c in cues], dtype=np.float64)
        self.end = np.asarray([c[1] for c in cues], dtype=np.float64)
        self.tol = float(tol)

    def lookup(self, times):
        """times [n] (seconds) -> int64 [n]: index of the cue shown at each time, -1 for none."""
        t = np.atleast_1d(np.asarray(times, dtype=np.float64))
        if not self.text:
            return np.full(t.shape, -1, dtype=np.int64)
        cand = np.searchsorted(self.start, t + self.tol, side="right") - 1
        c = np.maximum(cand, 0)
        on = (cand >= 0) & (self.start[c] - self.tol <= t) & (t < self.end[c] + self.tol)
        prev = np.maximum(c - 1, 0)
        handed = (cand >= 0) & ~on & (t < self.start[c]) & (self.start[prev] - self.tol <= t) & (t < self.end[prev] + self.tol)
        return np.where(on, c, np.where(handed, prev, -1)).astype(np.int64)

    def texts(self, times):
        return [self.text[i] if i >= 0 else None for i in self.lookup(times)]

    def at(self, t):
        return self.texts([t])[0]


def _text_size(draw, font, text):
    x0, y0, x1, y1 = draw.textbbox((0, 0), text, font=font, stroke_width=0)
    return x1 - x0, y1 - y0


def wrap_words(draw, font, text, max_width):
    """Lay text out in lines no wider than max_width pixels: explicit line breaks are kept (an empty paragraph is an empty
    line), words are never split, a line takes words while it still fits (the greedy rule of vision_process.py:221-240)."""
    fits = lambda words: _text_size(draw, font, " ".join(words))[0] <= max_width
    lines = []
    for paragraph in text.split("\n"):
        row = []
        for word in paragraph.split():
            if row and not fits(row + [word]):
                lines.append(" ".join(row))
                row = []
            row.append(word)
        lines.append(" ".join(row))
    return lines


def draw_subtitle(frame, text):
    """Wrapped white text on a 55 % black box, bottom centre; font 6 % of the short side clamped to [20, 96] px, wrap at 90 % of
    the width, margin 25 % of the font (>= 6), line gap 20 % of the font (vision_process.py:45-52, :251-309)."""
    if not text:
        return frame
    from PIL import Image, ImageDraw
    img = Image.fromarray(np.ascontiguousarray(frame), mode="RGB").convert("RGBA")
    px = max(20, min(96, int(round(min(frame.shape[:2]) * 0.06))))
    font, margin, stroke, gap = _font(px), max(6, int(px * 0.25)), max(1, px // 12), int(0.2 * px)
    W, H = img.size
    draw = ImageDraw.Draw(img)
    lines = wrap_words(draw, font, text, int(W * 0.90))
    sizes = [_text_size(draw, font, ln if ln else " ") for ln in lines]
    text_h = sum(h for _, h in sizes) + gap * max(0, len(lines) - 1)
    text_w = max((w for w, _ in sizes), default=0)
    left = max(0, (W - text_w) // 2 - margin)
    right = min(W, left + text_w + 2 * margin)
    bottom = H - margin
    top = max(0, bottom - text_h - 2 * margin)
    box = Image.new("RGBA", img.size, (0, 0, 0, 0))
    ImageDraw.Draw(box).rectangle([left, top, right, bottom], fill=(0, 0, 0, int(255 * 0.55)))
    img = Image.alpha_composite(img, box)
    draw = ImageDraw.Draw(img)
    y = top + margin
    for ln, (w, h) in zip(lines, sizes):
        draw.text(((W - w) // 2, y), ln, fill=(255, 255, 255), font=font, stroke_width=stroke, stroke_fill=(0, 0, 0))
        y += h + gap
    return np.asarray(img.convert("RGB"))


# ---------------------------------------------------------------------------------------------- one video element
def read_video(ele, clip=None):
    """One {"type": "video", ...} element -> (uint8 frames [n,3,H,W] with overlays, sample_fps, (H', W') to resize to).
    The decode-backend functions of the reference (vision_process.py:608-664 and its siblings) share this body."""
    path = ele["video"]
    clip = clip if clip is not None else open_clip(path)
    total, fps = len(clip), float(clip.fps)
    first, last, total = frame_range(ele, total, fps)
    n = hf.smart_nframes(ele, total_frames=total, video_fps=fps)
    idx = hf.frame_indices(total, n, start=first, end=last).tolist()
    fr = np.array(clip.get(idx), dtype=np.uint8)                       # [n,H,W,3], writable
    if _env_flag("WRITE_TIMESTAMPS_ON_FRAMES"):
        for k, i in enumerate(idx):
            fr[k] = draw_timestamp(fr[k], format_hhmmss(i / fps))       # absolute source time of the frame
    if _env_flag("WRITE_SUBTITLES_ON_FRAMES"):
        srt = ele["subtitles"] if "subtitles" in ele else (os.path.splitext(path)[0] + ".srt")
        if srt is not None and os.path.exists(srt):
            try:
                shown = SubtitleTrack(parse_srt(str(srt))).texts(np.asarray(idx, dtype=np.float64) / fps)
                for k, text in enumerate(shown):
                    if text:
                        fr[k] = draw_subtitle(fr[k], text)
            except Exception as e:                                      # a bad subtitle file must not stop training
                print(f"[frontend] subtitle overlay skipped for {path}: {e}")
    frames = torch.from_numpy(fr).permute(0, 3, 1, 2).contiguous()
    dims = hf.video_resize_dims(n, fr.shape[1], fr.shape[2], ele)
    return frames, hf.sample_fps(n, total, fps), dims


# ---------------------------------------------------------------------------------------------- still images
PROCESSOR_MIN_PIXELS, PROCESSOR_MAX_PIXELS = 56 * 56, 28 * 28 * 16384      # Qwen2.5-VL preprocessor_config.json (min / max_pixels)


def processor_pixel_range(model_dir):
    """(min_pixels, max_pixels) of the checkpoint's image processor: `preprocessor_config.json` beside the weights (keys
    min_pixels / max_pixels, or size.shortest_edge / size.longest_edge in newer files); the Qwen2.5-VL values without one."""
    lo, hi = PROCESSOR_MIN_PIXELS, PROCESSOR_MAX_PIXELS
    path = os.path.join(model_dir, "preprocessor_config.json") if model_dir else None
    if path and os.path.exists(path):
        with open(path) as f:
            pc = json.load(f)
        size = pc.get("size") or {}
        lo = int(pc.get("min_pixels") or size.get("shortest_edge") or lo)
        hi = int(pc.get("max_pixels") or size.get("longest_edge") or hi)
    return lo, hi


def open_image(src):
    """PIL image from what an image element may hold (vision_process.py:374-400): a PIL image, a path, file://, http(s)://
    or a data:image;base64 URI."""
    from PIL import Image
    if isinstance(src, Image.Image):
        return src
    if isinstance(src, np.ndarray):
        return Image.fromarray(src)
    if src.startswith("http://") or src.startswith("https://"):
        import io
        import urllib.request
        with urllib.request.urlopen(src, timeout=10) as r:
            return Image.open(io.BytesIO(r.read())).copy()
    if src.startswith("data:image") and "base64," in src:
        import base64
        import io
        return Image.open(io.BytesIO(base64.b64decode(src.split("base64,", 1)[1]))).copy()
    return Image.open(src[7:] if src.startswith("file://") else src)


def to_rgb(img):
    """alpha composited on white, everything else converted (vision_process.py:365-371)"""
    from PIL import Image
    if img.mode == "RGBA":
        bg = Image.new("RGB", img.size, (255, 255, 255))
        bg.paste(img, mask=img.split()[3])
        return bg
    return img.convert("RGB")


def read_image(ele, processor_pixels=(PROCESSOR_MIN_PIXELS, PROCESSOR_MAX_PIXELS)):
    """One still-image element -> uint8 frame [1,3,H',W'] ready for the device (rescale + normalise + patchify there; the single
    frame fills both temporal slots of its patches, as HF's image processor repeats it).  The reference's two host stages:
    `fetch_image` (vision_process.py:401-420: RGB, smart_resize under the element's min / max pixels -- or the given
    resized_height / resized_width snapped to 28 -- then PIL's resize with its default filter) and the HF image processor's own
    smart_resize under the PROCESSOR's pixel range (a no-op for an image that is already inside it)."""
    from PIL import Image
    img = to_rgb(open_image(ele["image"] if "image" in ele else ele["image_url"]))
    if "resized_height" in ele and "resized_width" in ele:
        h, w = hf.smart_resize(ele["resized_height"], ele["resized_width"], factor=hf.IMAGE_FACTOR)
    else:
        w0, h0 = img.size
        h, w = hf.smart_resize(h0, w0, factor=hf.IMAGE_FACTOR, min_pixels=ele.get("min_pixels", hf.MIN_PIXELS),
                               max_pixels=ele.get("max_pixels", hf.MAX_PIXELS))
    img = img.resize((w, h))
    h2, w2 = hf.smart_resize(h, w, factor=hf.IMAGE_FACTOR, min_pixels=processor_pixels[0], max_pixels=processor_pixels[1])
    if (h2, w2) != (h, w):
        img = img.resize((w2, h2), resample=Image.BICUBIC)
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).unsqueeze(0).contiguous()


# ---------------------------------------------------------------------------------------------- chat layout + tokenisers
def render_chat(messages, add_generation_prompt=True):
    """The Qwen2.5-VL chat layout for the message shapes the dataset builds (system string + user list of media / text
    elements): what `maybe_apply_chat_template(example, processor)` returns for them (TRL)."""
    out = []
    if messages and messages[0]["role"] != "system":
        out.append(f"{IM_START}system\nYou are a helpful assistant.{IM_END}\n")
    for m in messages:
        out.append(f"{IM_START}{m['role']}\n")
        if isinstance(m["content"], str):
            out.append(m["content"])
        else:
            for c in m["content"]:
                if c.get("type") == "image" or "image" in c or "image_url" in c:
                    out.append(VISION_START + IMAGE_PAD + VISION_END)
                elif c.get("type") == "video" or "video" in c:
                    out.append(VISION_START + VIDEO_PAD + VISION_END)
                elif "text" in c:
                    out.append(c["text"])
        out.append(IM_END + "\n")
    if add_generation_prompt:
        out.append(f"{IM_START}assistant\n")
    return "".join(out)


class ByteTokenizer:
    """Offline stand-in tokeniser: UTF-8 bytes are ids 0..255, the chat / vision marker strings map to the model config's
    special ids (im_start gets a spare id).  Lossless on text, total on ids (unknown ids decode to nothing)."""

    def __init__(self, cfg):
        spare = [i for i in range(cfg.text.vocab_size - 1, 255, -1)
                 if i not in (cfg.image_token_id, cfg.video_token_id, cfg.vision_start_token_id, cfg.vision_end_token_id,
                              cfg.eos_token_id, cfg.pad_token_id)]
        self.special = {IM_START: spare[0], IM_END: cfg.eos_token_id, VISION_START: cfg.vision_start_token_id,
                        VISION_END: cfg.vision_end_token_id, VIDEO_PAD: cfg.video_token_id, IMAGE_PAD: cfg.image_token_id,
                        END_OF_TEXT: cfg.pad_token_id}
        self.by_id = {v: k for k, v in self.special.items()}
        self._split = re.compile("(" + "|".join(re.escape(s) for s in self.special) + ")")
        self.eos_token_id, self.pad_token_id = cfg.eos_token_id, cfg.pad_token_id

    def encode(self, text):
        ids = []
        for piece in self._split.split(text):
            if piece in self.special:
                ids.append(self.special[piece])
            elif piece:
                ids.extend(piece.encode("utf-8"))
        return ids

    def decode(self, ids, skip_special_tokens=True):
        out, run = [], bytearray()
        for i in (int(x) for x in ids):
            if i < 256:
                run.append(i)
                continue
            if run:
                out.append(run.decode("utf-8", errors="replace")); run = bytearray()
            if i in self.by_id and not skip_special_tokens:
                out.append(self.by_id[i])
        if run:
            out.append(run.decode("utf-8", errors="replace"))
        return "".join(out)


class HFTokenizer:
    """`tokenizer.json` of the model directory through the `tokenizers` runtime (no network, nothing executed from the file)."""

    def __init__(self, model_dir, cfg):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(os.path.join(model_dir, "tokenizer.json"))
        eos = self.tk.token_to_id(IM_END)
        pad = self.tk.token_to_id(END_OF_TEXT)
        self.eos_token_id = cfg.eos_token_id if eos is None else eos
        self.pad_token_id = cfg.pad_token_id if pad is None else pad

    def encode(self, text):
        return self.tk.encode(text, add_special_tokens=False).ids

    def decode(self, ids, skip_special_tokens=True):
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens)


def load_tokenizer(model_dir, cfg):
    if model_dir and os.path.exists(os.path.join(model_dir, "tokenizer.json")):
        return HFTokenizer(model_dir, cfg)
    return ByteTokenizer(cfg)


# ---------------------------------------------------------------------------------------------- the processing function
class ProcessingFrontEnd:
    """processing_fn(example) for QwenGRPOTrainer: dataset example -> prompt ids + uint8 clips + decode()."""

    def __init__(self, cfg, tokenizer, clip_opener=open_clip, model_dir=None):
        self.cfg, self.tok, self.open = cfg, tokenizer, clip_opener
        self.image_pixels = processor_pixel_range(model_dir)      # the HF image processor's own resize range (second image stage)

    def video_tokens(self, n_frames, h, w):
        v = self.cfg.vision
        t = (n_frames + v.temporal_patch_size - 1) // v.temporal_patch_size
        return t * (h // v.patch_size) * (w // v.patch_size) // v.spatial_merge_size ** 2

    def __call__(self, example):
        prompt = example["prompt"]
        videos, secs, counts, pads = [], [], [], []         # the prompt's vision items in prompt order: clips and still images
        for m in prompt:
            if not isinstance(m["content"], list):
                continue
            for c in m["content"]:
                if c.get("type") == "image" or "image" in c or "image_url" in c:
                    frame = read_image(c, self.image_pixels)  # [1,3,H',W'], already at its final size (PIL, as the reference)
                    videos.append((frame.pin_memory() if torch.cuda.is_available() else frame, None))
                    secs.append(None)                        # None = still image (HF has second_per_grid_ts for videos only)
                    counts.append(self.video_tokens(1, int(frame.shape[2]), int(frame.shape[3])))
                    pads.append(self.cfg.image_token_id)
                elif c.get("type") == "video" or "video" in c:
                    frames, sfps, dims = read_video(c, self.open(c["video"]))
                    videos.append((frames.pin_memory() if torch.cuda.is_available() else frames, dims))
                    secs.append(hp.second_per_grid_t(self.cfg.vision.temporal_patch_size, sfps))
                    counts.append(self.video_tokens(frames.shape[0], *dims))
                    pads.append(self.cfg.video_token_id)
        text = render_chat(prompt, add_generation_prompt=True)
        pieces = re.split("(" + re.escape(VIDEO_PAD) + "|" + re.escape(IMAGE_PAD) + ")", text)
        marks = [pc for pc in pieces[1::2]]
        want = [VIDEO_PAD if p_ == self.cfg.video_token_id else IMAGE_PAD for p_ in pads]
        if marks != want:
            raise ValueError(f"{marks.count(VIDEO_PAD)} video placeholders and {marks.count(IMAGE_PAD)} image placeholders in the prompt "
                             f"text but {want.count(VIDEO_PAD)} video and {want.count(IMAGE_PAD)} image elements (or in another order)")
        ids, k = [], 0
        for j, piece in enumerate(pieces):
            if j % 2 == 0:
                ids.extend(self.tok.encode(piece))
            else:
                ids.extend([pads[k]] * counts[k])            # one pad token per merged patch (HF processing_qwen2_5_vl.py:138-154)
                k += 1
        tok = self.tok

        def decode(completion_ids):
            return [tok.decode(row, skip_special_tokens=True) for row in np.asarray(torch.as_tensor(completion_ids).cpu())]

        return {"input_ids": np.asarray(ids, dtype=np.int64), "videos": videos, "second_per_grid_ts": secs or None,
                "decode": decode, "prompt_text": text}
