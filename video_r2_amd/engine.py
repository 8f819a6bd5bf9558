"""MI355X rollout engine: ViT encode -> LM prefill (once per prompt) -> shared-prefix decode of G completions ->
teacher-forced per-token log-probs.  Every arithmetic step is a libvr2.so kernel (include/vr2.h); torch is used
only to own device memory and streams.

Replaces, for ONE prompt group at a time (the DP unit):
  * ``model.generate(**prompt_inputs, generation_config)``   /root/reference/train/src/trainer/grpo_trainer.py:783-786
  * ``_get_per_token_logps(model, ids, mask, C, **mm)``        grpo_trainer.py:723-741
Differences from the reference's execution (results identical): the ViT and the prompt prefill run once
per prompt (not once per completion / per pass), the prompt KV is stored once and shared by the G decodes, and the
frozen vision embeddings are reused by the policy and reference scoring passes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import os

import numpy as np
import torch

from . import lib
from .config import ModelConfig
from .host import rope_index as ri
from .host import vision_plan as vp
from .weights import Params, pad64

EPI_BF16, EPI_F32, EPI_ADD_F32, EPI_GELU_BF16, EPI_SWIGLU_BF16 = 0, 1, 2, 3, 4
DECODE_CHUNK = int(os.environ.get('VR2_DECODE_CHUNK', '0'))     # suffix keys per decode-attention workgroup; 0 = pick per call
PREFIX_CHUNK = int(os.environ.get('VR2_PREFIX_CHUNK', '0'))     # prompt keys per shared-prefix workgroup; 0 = pick per call


def _i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int32))).to(device)


def _on_stream(fn):
    """Run a public Engine method with the engine's HIP stream as torch's current stream (so every allocation, H2D copy
    and tensor op it issues is ordered with the kernels), ordered after the caller's stream on entry and before it on exit."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        outer = torch.cuda.current_stream(self.device)
        if outer == self.stream:
            return fn(self, *args, **kwargs)
        self.stream.wait_stream(outer)
        with torch.cuda.stream(self.stream):
            out = fn(self, *args, **kwargs)
        outer.wait_stream(self.stream)
        return out
    return wrapper


@dataclass
class PromptState:
    """Everything the decode / scoring passes need about one prefilled prompt."""
    n_prompt: int
    rope_delta: int
    prefix_k: torch.Tensor      # [L][S][nkv*hd] bf16
    prefix_v: torch.Tensor
    last_hidden: torch.Tensor   # [1][H] f32 residual stream of the last prompt token (pre final norm)
    last_logits: torch.Tensor   # [1][V] f32
    input_ids: object = None    # host ints [S]: the prompt's token ids (repetition penalty: HF applies it to prompt + generated ids)


class Engine:
    def __init__(self, cfg: ModelConfig, vision: Params, text: Params, device="cuda:0", stream=None):
        if not torch.cuda.is_available():
            raise lib.Vr2Error("video_r2_amd.Engine needs an MI355X (no CPU fallback)")
        lib.load()
        lib.gemm_workspace(device)      # slabs of the GEMMs' deterministic split-K tails (vr2_set_gemm_workspace)
        self.cfg = cfg
        self.vision = vision
        self.text = text
        self.device = torch.device(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self._ws = {}
        self._ws_epoch = 0
        self._plans = {}
        t, v = cfg.text, cfg.vision
        hd = t.head_dim
        inv = 1.0 / (t.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        self.inv_freq = inv.to(self.device)
        sec = []
        for i, n in enumerate(list(t.mrope_section)):
            sec += [i] * n
        assert len(sec) == hd // 2, "mrope_section must sum to head_dim/2"
        self.section = _i32(sec, self.device)
        vdim = v.head_dim // 2
        self.vis_inv_freq = (1.0 / (10000.0 ** (torch.arange(0, vdim, 2, dtype=torch.float32) / vdim))).to(self.device)
        self._graphs = {}
        self.decode_fp8 = False      # decode GEMVs stream per-channel e4m3 weight copies (weights.add_fp8_decode_copies; config 5)
        self.decode_fp8_max_rows = 128
        self.decode_fp8_act = True   # ... and, above 16 rows, fp8 ACTIVATIONS on the fp8 MFMA (vr2_quant_rows_fp8 + vr2_gemv_a8w8)
        self.gemm_fp8 = False        # config 5, the no-grad LM passes (prompt prefill, teacher-forced scoring): per-row e4m3
                                     # activations x the same per-channel e4m3 weight copies on the block-scaled fp8 MFMA
                                     # (vr2_gemm_a8w8); the training forward / backward stay bf16

    # ------------------------------------------------------------------ plumbing
    @property
    def s(self):
        return self.stream.cuda_stream

    def buf(self, name, shape, dtype, zero=False):
        """Named workspace with CAPACITY semantics: one allocation per name that only ever grows along its row dimension
        (dim 0 for 1-/2-D, dim 1 for [layers|splits][rows][cols], dim 2 for [groups][layers][rows][cols]) and is handed out
        as a view of the requested size -- prompts of different lengths re-use the same memory (and the same captured
        decode graph) instead of leaking one buffer set per length.  zero=True buffers are zero-filled at (re)allocation;
        kernels never write their padding columns."""
        shape = tuple(int(x) for x in shape)
        gd = {1: 0, 2: 0, 3: 1, 4: 2}[len(shape)]
        key = (name, dtype, shape[:gd] + shape[gd + 1:])
        b = self._ws.get(key)
        need = shape[gd]
        if b is None or b.shape[gd] < need:
            cap = need if need <= 256 else (need + need // 8 + 127) // 128 * 128       # 12.5 % headroom on long dimensions
            full = shape[:gd] + (cap,) + shape[gd + 1:]
            make = torch.zeros if zero else torch.empty
            with torch.cuda.stream(self.stream):
                try:
                    b = make(full, dtype=dtype, device=self.device)
                except torch.OutOfMemoryError:
                    if torch.cuda.is_current_stream_capturing():
                        raise                    # no synchronize / cache flush inside a stream capture: the caller sizes its buffers before it
                    # the buffer this one replaces and the caching allocator's free blocks may be all that is in the way.  The old
                    # block goes back to the allocator NOW: captured decode graphs and buffer dicts that hold its address are stale
                    # from this moment, whether or not the second attempt succeeds
                    if self._ws.pop(key, None) is not None:
                        self._ws_epoch += 1
                        self._drop_graphs()
                    b = None
                    self.stream.synchronize()
                    torch.cuda.empty_cache()
                    try:
                        b = make(full, dtype=dtype, device=self.device)
                    except torch.OutOfMemoryError as e:
                        free, total = torch.cuda.mem_get_info(self.device)
                        gib = float(np.prod(full)) * torch.empty((), dtype=dtype).element_size() / 2 ** 30
                        raise torch.OutOfMemoryError(
                            f"workspace '{name}' {full} {dtype} ({gib:.1f} GiB) does not fit: {free / 2 ** 30:.1f} GiB free of "
                            f"{total / 2 ** 30:.0f} GiB.  The saved activations of a training pass grow with (prompt rows + "
                            "num_generations x completion rows) x layers; prefix caches and decode buffers with the prompt groups "
                            "decoded together: lower --max_completion_length / --num_generations, the frame budget (FPS_MAX_FRAMES, "
                            "--video_max_pixels) or --per_device_train_batch_size.") from e
            self._ws[key] = b
            self._ws_epoch += 1                                                         # captured graphs hold old addresses
        return b.narrow(gd, 0, need)

    def sync(self):
        self.stream.synchronize()

    def _drop_graphs(self):
        """destroy every captured decode graph (a workspace moved or was freed: the graphs hold old addresses)"""
        for ex_ in self._graphs.values():
            lib.call("vr2_graph_destroy", ex_)
        self._graphs.clear()
        self._graph_epoch = self._ws_epoch

    def _gemm(self, A, P: Params, wname, bias, C, M, epi):
        n, kp = P.shape[wname]
        lib.call("vr2_gemm_bf16", A, A.stride(0), P[wname], bias, C, C.stride(0), M, n, kp, epi, self.s)

    def _lm_f8(self, P: Params):
        """the fp8 GEMM path of _lm_rows / _logits_rows applies: mode on, P carries fp8 copies, every K a multiple of 128"""
        if not self.gemm_fp8:
            return False
        f8 = getattr(P, "f8", None)
        if not f8:
            raise lib.Vr2Error("gemm_fp8 is on but the parameters have no fp8 copies: call weights.add_fp8_decode_copies(params) first")
        t = self.cfg.text
        return all(k % 128 == 0 for k in (t.hidden_size, t.intermediate_size, t.num_attention_heads * t.head_dim)) and t.hidden_size <= 4096

    def _gemm8(self, q8, xs, P: Params, wname, bias, C, M, epi):
        n, kp = P.shape[wname]
        w8, ws = P.f8[wname]
        lib.call("vr2_gemm_a8w8", q8, q8.stride(0), xs, w8, ws, bias, C, C.stride(0), M, n, kp, epi, self.s)

    def _quant_rows(self, x, M, K, tag):
        """bf16 rows -> (e4m3 bytes, per-row scales) in a workspace of the calling pass (not the decode graph's)"""
        t = self.cfg.text
        width = 128 * ((max(t.intermediate_size, t.hidden_size, t.num_attention_heads * t.head_dim) + 127) // 128)
        q8, xs = self.buf(tag + ".q8", (M, width), torch.uint8), self.buf(tag + ".q8s", (M,), torch.float32)
        if x is not None:
            lib.call("vr2_quant_rows_fp8", x, x.stride(0), M, K, q8, q8.stride(0), xs, self.s)
        return q8, xs

    def _f8(self, P: Params, wname, rows=1):
        """(packed fp8 weights, scales) when fp8 decode weights are enabled and P carries them (weights.add_fp8_decode_copies).
        Measured decode speed-up over bf16 weights: 1.24-1.33x at 8-64 rows, 1.12x at 128 rows (where the activations' L2 -> LDS
        traffic, not the weight bytes, is most of the load path).  Above decode_fp8_max_rows the bf16 kernels are used."""
        if not self.decode_fp8 or rows > (256 if self.decode_fp8_act else self.decode_fp8_max_rows):
            return None
        f8 = getattr(P, "f8", None)
        if not f8 or wname not in f8:
            raise lib.Vr2Error(f"decode_fp8 is on but {wname} has no fp8 copy: call weights.add_fp8_decode_copies(params) first")
        return f8[wname]

    def _quant(self, x, G, K):
        """bf16 activation rows -> (e4m3 bytes in the fragment order of vr2_gemv_a8w8, per-row scales); one shared workspace"""
        q8, xs = self._qbuf(G)
        lib.call("vr2_quant_rows_fp8", x, x.stride(0), G, K, q8, q8.stride(0), xs, self.s)
        return q8, xs

    def _a8w8(self, P: Params, wname, G):
        """fp8 activations x fp8 weights on the fp8 MFMA: above 16 rows, K a multiple of 128, when the mode is on"""
        return self.decode_fp8 and self.decode_fp8_act and 16 < G <= 256 and P.shape[wname][1] % 128 == 0

    def _qbuf(self, G):
        t = self.cfg.text
        width = 128 * ((max(t.intermediate_size, t.hidden_size, t.num_attention_heads * t.head_dim) + 127) // 128)
        return self.buf("d.q8", (G, width), torch.uint8), self.buf("d.q8s", (G,), torch.float32)

    def _gemv(self, x, P: Params, wname, G, out, nsplit, prequant=None):
        n, kp = P.shape[wname]
        f8 = self._f8(P, wname, G)
        if f8 is not None and self._a8w8(P, wname, G) and kp // 128 >= nsplit:
            q8, xs = prequant if prequant is not None else self._quant(x, G, kp)
            lib.call("vr2_gemv_a8w8", q8, q8.stride(0), xs, G, f8[0], f8[1], n, kp, out, out.stride(0) if nsplit > 1 else 0,
                     out.stride(-2), nsplit, self.s)
            return
        if f8 is not None and G <= self.decode_fp8_max_rows:      # weight-only fp8 (bf16 activations): up to 128 rows
            lib.call("vr2_gemv_w8", x, x.stride(0), G, f8[0], f8[1], n, kp, out, out.stride(0) if nsplit > 1 else 0,
                     out.stride(-2), nsplit, self.s)
            return
        lib.call("vr2_gemv_bf16", x, x.stride(0), G, P[wname], n, kp, out, out.stride(0) if nsplit > 1 else 0,
                 out.stride(-2), nsplit, self.s)

    def _norm(self, x, w, y, M, H, eps, partial=None, nsplit=0, bias=None, q8=None):
        """q8 = (bytes, scales): also emit the row's e4m3 quantisation for the fp8-activation GEMV that follows (decode, config 5)"""
        if q8 is not None:
            ps = (partial, partial.stride(0), partial.stride(1), nsplit, bias) if partial is not None else (None, 0, 0, 0, None)
            lib.call("vr2_add_rmsnorm_q8", x, x.stride(0), *ps, w, y, y.stride(0), M, H, eps, q8[0], q8[0].stride(0), q8[1], self.s)
            return
        if partial is None:
            lib.call("vr2_add_rmsnorm", x, x.stride(0), None, 0, 0, 0, None, w, y, 0 if y is None else y.stride(0), M, H, eps, self.s)
        else:
            lib.call("vr2_add_rmsnorm", x, x.stride(0), partial, partial.stride(0), partial.stride(1), nsplit, bias, w, y,
                     0 if y is None else y.stride(0), M, H, eps, self.s)

    # ------------------------------------------------------------------ vision tower (K1-K7)
    def vision_plan(self, grid_thw):
        key = tuple(int(x) for x in np.asarray(grid_thw).reshape(-1))
        pl = self._plans.get(key)
        if pl is None:
            v = self.cfg.vision
            pl = vp.make_plan(np.asarray(grid_thw).reshape(-1, 3), v.spatial_merge_size, v.window_size, v.patch_size)
            pl.d_gather = _i32(pl.gather_rows, self.device)
            pl.d_pos = _i32(pl.pos_hw, self.device)
            pl.d_work_window = _i32(pl.work_window, self.device)
            pl.d_work_full = _i32(pl.work_full, self.device)
            self._plans[key] = pl
        return pl

    @_on_stream
    def resize_frames(self, frames_u8, out_h, out_w):
        """uint8 [N,3,H,W] -> uint8 [N,3,out_h,out_w] on the device: antialiased bicubic, width pass then height pass over a
        uint8 intermediate -- bit for bit the reference's torchvision resize (vision_process.py:802-807)."""
        from .host import frames as hf
        fr = torch.as_tensor(frames_u8)
        if fr.dtype != torch.uint8 or fr.dim() != 4:
            raise ValueError(f"resize_frames wants uint8 [N,C,H,W], got {fr.dtype} {tuple(fr.shape)}")
        fr = fr.to(self.device).contiguous()
        n, c, hh, ww = (int(a) for a in fr.shape)
        cur = fr
        for vertical, (n_in, n_out) in ((0, (ww, int(out_w))), (1, (hh, int(out_h)))):
            if n_in == n_out:
                continue
            key = ("taps", n_in, n_out)
            tp = self._plans.get(key)
            if tp is None:
                first, count, wt, bits = hf.resize_taps(n_in, n_out)
                tp = (first.to(self.device), count.to(self.device), wt.to(self.device).contiguous(), int(bits))
                self._plans[key] = tp
            rows_in, cols_in = int(cur.shape[2]), int(cur.shape[3])
            rows_out, cols_out = (n_out, cols_in) if vertical else (rows_in, n_out)
            dst = torch.empty((n, c, rows_out, cols_out), dtype=torch.uint8, device=self.device)
            lib.call("vr2_resize_axis_u8", cur, dst, tp[0], tp[1], tp[2], tp[2].shape[1], tp[3], n * c, rows_in, cols_in,
                     rows_out, cols_out, vertical, self.s)
            cur = dst
        return cur

    @_on_stream
    def encode_frames(self, frames_u8, rescale_factor=1.0 / 255.0, resize_to=None):
        """uint8 frames [N,3,H,W] of ONE clip (H, W multiples of 28 after the optional resize_to=(H', W'), which runs
        resize_frames first) -> (video tokens, plan, grid) as encode_video.
        Rescale + normalise + patchify + window gather + bf16 cast run in one kernel on the device (vr2_patchify_u8): the
        values are bit-identical to host.processor.normalize_frames + patchify followed by the cast (HF video processor)."""
        import ctypes
        from .host import processor as hp
        v = self.cfg.vision
        fr = torch.as_tensor(frames_u8)
        if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[1] != 3:
            raise ValueError(f"encode_frames wants uint8 [N,3,H,W], got {fr.dtype} {tuple(fr.shape)}")
        if resize_to is not None:
            fr = self.resize_frames(fr, resize_to[0], resize_to[1])
        n, _, hh, ww = (int(a) for a in fr.shape)
        unit = v.patch_size * v.spatial_merge_size
        if hh % unit or ww % unit:
            raise ValueError(f"frame size {hh}x{ww} is not a multiple of {unit} (smart_resize first)")
        grid = ((n + v.temporal_patch_size - 1) // v.temporal_patch_size, hh // v.patch_size, ww // v.patch_size)
        pl = self.vision_plan([grid])
        P, Kp = pl.n_patches, pad64(v.patch_dim)
        fr = fr.to(self.device).contiguous()
        mean = (torch.tensor(hp.CLIP_MEAN, dtype=torch.float32) * (1.0 / rescale_factor)).tolist()   # same fp32 constants as the host path
        std = (torch.tensor(hp.CLIP_STD, dtype=torch.float32) * (1.0 / rescale_factor)).tolist()
        a0 = self.buf("v.px", (P, Kp), torch.bfloat16)
        lib.call("vr2_patchify_u8", fr, n, hh, ww, pl.d_gather, (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), a0,
                 a0.stride(0), P, v.patch_size, v.temporal_patch_size, v.spatial_merge_size, Kp, self.s)
        emb = self._vision_blocks(pl, a0)
        return emb, pl, grid

    @_on_stream
    def encode_video(self, pixel_values, grid_thw):
        """pixel_values f32 [P, C*tp*ps*ps] (device) -> merged video tokens f32 [P/4, out_hidden] in WINDOW order plus
        the plan (plan.window_index gives each row's natural token index)."""
        v = self.cfg.vision
        pl = self.vision_plan(grid_thw)
        P = pl.n_patches
        if pixel_values.shape[0] != P or pixel_values.shape[1] != v.patch_dim:
            raise ValueError(f"pixel_values shape {tuple(pixel_values.shape)} does not match grid {grid_thw} ({P} x {v.patch_dim})")
        Kp = pad64(v.patch_dim)
        px = pixel_values.to(device=self.device, dtype=torch.float32).contiguous()
        a0 = self.buf("v.px", (P, Kp), torch.bfloat16)
        lib.call("vr2_cast_rows_f32_bf16", px, px.stride(0), pl.d_gather, a0, a0.stride(0), P, v.patch_dim, Kp, self.s)
        return self._vision_blocks(pl, a0), pl

    @_on_stream
    def encode_videos(self, clips, per_prompt=None, vision=None, rows_only=False):
        """Several clips through ONE pass of the vision tower (M = all their patches: the N = 1280 / 3420 projections then run
        on full 256-row tiles instead of one 4096-patch clip at a time).  clips: list of dict(pixel_values=f32 [P_i, C*tp*ps*ps],
        grid=(t,h,w)) or dict(frames_u8=uint8 [n,3,H,W], resize_to=(H', W') | None).  per_prompt: how many consecutive clips
        each prompt owns (default: one each; 0 = a prompt without video -> (None, None)).  -> per prompt (video tokens f32
        [Tv, out_hidden] in the window order of the plan over ITS clips, that plan); views of one buffer owned by the caller.
        vision: run another tower than the engine's own (the frozen reference model's when the policy's vision side trains).
        rows_only: stop in front of the tower -> (patch rows bf16 [P][Kpad] in window order (a copy), the plan over ALL clips): the
        input of train_vision.VisionTrainer.forward, which runs the tower with saved activations."""
        import ctypes
        from .host import processor as hp
        v = self.cfg.vision
        Kp = pad64(v.patch_dim)
        metas = []
        for c in clips:
            if c.get("frames_u8") is not None:
                fr = torch.as_tensor(c["frames_u8"])
                if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[1] != 3:
                    raise ValueError(f"encode_videos wants uint8 [N,3,H,W] frames, got {fr.dtype} {tuple(fr.shape)}")
                if c.get("resize_to") is not None:
                    fr = self.resize_frames(fr, c["resize_to"][0], c["resize_to"][1])
                n, _, hh, ww = (int(a) for a in fr.shape)
                unit = v.patch_size * v.spatial_merge_size
                if hh % unit or ww % unit:
                    raise ValueError(f"frame size {hh}x{ww} is not a multiple of {unit} (smart_resize first)")
                grid = ((n + v.temporal_patch_size - 1) // v.temporal_patch_size, hh // v.patch_size, ww // v.patch_size)
                metas.append(("u8", fr.to(self.device).contiguous(), grid))
            else:
                grid = tuple(int(x) for x in np.asarray(c["grid"]).reshape(-1))
                px = c["pixel_values"]
                if px.shape[0] != grid[0] * grid[1] * grid[2] or px.shape[1] != v.patch_dim:
                    raise ValueError(f"pixel_values shape {tuple(px.shape)} does not match grid {grid}")
                metas.append(("px", px.to(device=self.device, dtype=torch.float32).contiguous(), grid))
        grids = [m[2] for m in metas]
        per_prompt = [1] * len(metas) if per_prompt is None else list(per_prompt)
        if sum(per_prompt) != len(metas):
            raise ValueError(f"per_prompt {per_prompt} does not add up to {len(metas)} clips")
        if not metas:
            return [(None, None)] * len(per_prompt)
        pl_all = self.vision_plan(grids)
        a0 = self.buf("v.px", (pl_all.n_patches, Kp), torch.bfloat16)
        singles, row = [], 0
        for kind, data, grid in metas:
            pl = self.vision_plan([grid])
            dst = a0[row:row + pl.n_patches]
            if kind == "u8":
                n, _, hh, ww = (int(a) for a in data.shape)
                mean = (torch.tensor(hp.CLIP_MEAN, dtype=torch.float32) * 255.0).tolist()
                std = (torch.tensor(hp.CLIP_STD, dtype=torch.float32) * 255.0).tolist()
                lib.call("vr2_patchify_u8", data, n, hh, ww, pl.d_gather, (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), dst,
                         dst.stride(0), pl.n_patches, v.patch_size, v.temporal_patch_size, v.spatial_merge_size, Kp, self.s)
            else:
                lib.call("vr2_cast_rows_f32_bf16", data, data.stride(0), pl.d_gather, dst, dst.stride(0), pl.n_patches, v.patch_dim, Kp, self.s)
            singles.append(pl)
            row += pl.n_patches
        if rows_only:
            return a0.clone(), pl_all
        out = self._vision_blocks(pl_all, a0, vision).clone()          # the combined window order is clip after clip
        res, tok, ci = [], 0, 0
        for k in per_prompt:
            if k == 0:
                res.append((None, None))
                continue
            pl = singles[ci] if k == 1 else self.vision_plan(grids[ci:ci + k])
            res.append((out[tok:tok + pl.n_tokens], pl))
            tok += pl.n_tokens
            ci += k
        return res

    def _vision_blocks(self, pl, a0, W=None):
        """patch-embed GEMM input a0 bf16 [P][Kpad] (window order) -> merged video tokens (K1-K7).  W: another vision parameter set
        than the engine's own (the frozen reference model's tower when the policy's trains)"""
        W = self.vision if W is None else W
        return self._vision_merger(W, self._vision_tower(W, pl, a0), pl.n_patches)

    def _vision_tower(self, W: Params, pl, a0):
        """patch embedding + the ViT blocks -> residual stream x f32 [P][hidden] (window order), before the merger"""
        v = self.cfg.vision
        P = pl.n_patches
        Hv, hd, nh = v.hidden_size, v.head_dim, v.num_heads
        Hp, Ip, Kp = pad64(Hv), pad64(v.intermediate_size), pad64(v.patch_dim)
        x = self.buf("v.x", (P, Hv), torch.float32)
        self._gemm(a0, W, "patch_w", None, x, P, EPI_F32)
        cs = self.buf("v.cos", (P, hd // 2), torch.float32)
        sn = self.buf("v.sin", (P, hd // 2), torch.float32)
        lib.call("vr2_vision_rope_table", pl.d_pos, self.vis_inv_freq, cs, sn, P, hd, self.s)
        xn = self.buf("v.xn", (P, Hp), torch.bfloat16, zero=True)
        qkv32 = self.buf("v.qkv32", (P, 3 * Hv), torch.float32)
        qkv = self.buf("v.qkv", (P, 3 * Hv), torch.bfloat16)
        att = self.buf("v.att", (P, Hp), torch.bfloat16, zero=True)
        act = self.buf("v.act", (P, Ip), torch.bfloat16, zero=True)
        scale = hd ** -0.5
        for i in range(v.depth):
            full = i in v.fullatt_block_indexes
            work = pl.d_work_full if full else pl.d_work_window
            self._norm(x, W[f"b{i}.norm1"], xn, P, Hv, 1e-6)
            self._gemm(xn, W, f"b{i}.qkv_w", W[f"b{i}.qkv_b"], qkv32, P, EPI_F32)
            kq = qkv[:, Hv:]
            vq = qkv[:, 2 * Hv:]
            lib.call("vr2_rope_qkv", qkv32, qkv32.stride(0), cs, sn, P, nh, nh, hd, qkv, qkv.stride(0), kq, vq, qkv.stride(0), 0, self.s)
            lib.call("vr2_attn_tok", work, work.shape[0], qkv, qkv.stride(0), kq, vq, kq, vq, qkv.stride(0), att,
                     att.stride(0), nh, 1, hd, scale, None, self.s)
            self._gemm(att, W, f"b{i}.proj_w", W[f"b{i}.proj_b"], x, P, EPI_ADD_F32)
            self._norm(x, W[f"b{i}.norm2"], xn, P, Hv, 1e-6)
            self._gemm(xn, W, f"b{i}.gu_w", W[f"b{i}.gu_b"], act, P, EPI_SWIGLU_BF16)
            self._gemm(act, W, f"b{i}.down_w", W[f"b{i}.down_b"], x, P, EPI_ADD_F32)
        return x

    def _vision_merger(self, W: Params, x, P):
        """ln_q + the 2-layer MLP over groups of merge_size^2 patches -> video tokens f32 [P/4][out_hidden]"""
        v = self.cfg.vision
        Hv = v.hidden_size
        unit = v.spatial_merge_size ** 2
        Tv = P // unit
        mh = Hv * unit
        mn = self.buf("v.mn", (P, Hv), torch.bfloat16)
        self._norm(x, W["ln_q"], mn, P, Hv, 1e-6)
        mnv = mn.view(Tv, mh)
        h1 = self.buf("v.h1", (Tv, mh), torch.bfloat16)
        self._gemm(mnv, W, "fc1_w", W["fc1_b"], h1, Tv, EPI_GELU_BF16)
        out = self.buf("v.out", (Tv, v.out_hidden_size), torch.float32)
        self._gemm(h1, W, "fc2_w", W["fc2_b"], out, Tv, EPI_F32)
        return out

    # ------------------------------------------------------------------ LM building blocks
    def _lm_rows(self, W: Params, x, T, cos, sin, work, kA, vA, k_own, v_own, own_row0, tag, seg=None):
        """All decoder layers over T rows (prefill or teacher-forced scoring).  k_own/v_own: per-layer [L][rows][nkv*hd]
        caches (prefill) or a single [rows][nkv*hd] scratch reused by every layer (scoring).  seg = (rows per prompt, slot
        stride in rows): batched prefill, the K/V rows of prompt p go to slot p (vr2_rope_qkv_seg) and k_own[l] is slot 0."""
        t = self.cfg.text
        H, hd, nq, nkv, I = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size
        nqkv = (nq + 2 * nkv) * hd
        xn = self.buf(tag + ".xn", (T, H), torch.bfloat16)
        qkv = self.buf(tag + ".qkv32", (T, nqkv), torch.float32)
        qb = self.buf(tag + ".q", (T, nq * hd), torch.bfloat16)
        att = self.buf(tag + ".att", (T, nq * hd), torch.bfloat16)
        act = self.buf(tag + ".act", (T, I), torch.bfloat16)
        scale = hd ** -0.5
        f8 = self._lm_f8(W)
        qx = self._quant_rows(None, T, 0, tag) if f8 else None       # (bytes, scales) workspace of this pass
        for l in range(t.num_hidden_layers):
            ko = k_own[l] if k_own.dim() == 3 else k_own
            vo = v_own[l] if v_own.dim() == 3 else v_own
            ka = ko if kA is None else kA[l]
            va = vo if vA is None else vA[l]
            self._norm(x, W[f"l{l}.ln1"], xn, T, H, t.rms_norm_eps, q8=qx)
            if f8:
                self._gemm8(*qx, W, f"l{l}.qkv_w", W[f"l{l}.qkv_b"], qkv, T, EPI_F32)
            else:
                self._gemm(xn, W, f"l{l}.qkv_w", W[f"l{l}.qkv_b"], qkv, T, EPI_F32)
            if seg is None:
                lib.call("vr2_rope_qkv", qkv, qkv.stride(0), cos, sin, T, nq, nkv, hd, qb, qb.stride(0), ko, vo, ko.stride(0), own_row0, self.s)
            else:
                lib.call("vr2_rope_qkv_seg", qkv, qkv.stride(0), cos, sin, T, nq, nkv, hd, qb, qb.stride(0), ko, vo, ko.stride(0),
                         seg[0], seg[1], self.s)
            lib.call("vr2_attn_tok", work, work.shape[0], qb, qb.stride(0), ka, va, ko, vo, ko.stride(0), att,
                     att.stride(0), nq, t.group, hd, scale, None, self.s)
            if f8:
                self._gemm8(*self._quant_rows(att, T, nq * hd, tag), W, f"l{l}.o_w", None, x, T, EPI_ADD_F32)
                self._norm(x, W[f"l{l}.ln2"], xn, T, H, t.rms_norm_eps, q8=qx)
                self._gemm8(*qx, W, f"l{l}.gu_w", None, act, T, EPI_SWIGLU_BF16)
                self._gemm8(*self._quant_rows(act, T, I, tag), W, f"l{l}.down_w", None, x, T, EPI_ADD_F32)
                continue
            self._gemm(att, W, f"l{l}.o_w", None, x, T, EPI_ADD_F32)
            self._norm(x, W[f"l{l}.ln2"], xn, T, H, t.rms_norm_eps)
            self._gemm(xn, W, f"l{l}.gu_w", None, act, T, EPI_SWIGLU_BF16)
            self._gemm(act, W, f"l{l}.down_w", None, x, T, EPI_ADD_F32)
        return x

    def _logits_rows(self, W: Params, x, T, out, xn=None, allow_f8=True):
        """final norm + lm_head over T rows -> fp32 logits in ``out``.  xn: bf16 [T][H] of the caller to receive the normed rows
        (the training forward keeps them for the lm_head weight gradient; it also passes allow_f8=False: config 5's fp8 GEMMs
        are for the no-grad passes only)."""
        t = self.cfg.text
        # capacity semantics like every workspace: ONE buffer for the chunked callers (scoring: sum(lens) % 1024 differs for almost
        # every group) and one for the few-row callers (prompt-last rows); a name per row count leaked a buffer per distinct T
        # and bumped _ws_epoch each time, which drops the captured decode graphs
        if xn is None:
            xn = self.buf("lm.hn" if T > 128 else "lm.hn.s", (T, t.hidden_size), torch.bfloat16)
        if allow_f8 and T > 16 and self._lm_f8(W) and "lm_head" in W.f8:
            qx = self._quant_rows(None, T, 0, "lm")
            self._norm(x, W["norm"], xn, T, t.hidden_size, t.rms_norm_eps, q8=qx)
            self._gemm8(*qx, W, "lm_head", None, out, T, EPI_F32)
            return out
        self._norm(x, W["norm"], xn, T, t.hidden_size, t.rms_norm_eps)
        self._gemm(xn, W, "lm_head", None, out, T, EPI_F32)
        return out

    # ------------------------------------------------------------------ prompt prefill (K8-K12)
    def vision_layout(self, ids, plan, second_per_grid_ts, embeds):
        """M-RoPE positions, rope delta and the prompt positions of the vision tokens for one unpadded prompt whose vision items
        (plan.grid_thw, in prompt order) are video clips and / or still images -- an item is a STILL IMAGE when its
        second_per_grid_ts entry is None (host/rope_index.vision_layout).  Raises HF's "... features and ... tokens do not match"."""
        cfg = self.cfg
        grids = [] if plan is None else plan.grid_thw
        pos, delta, vis_pos = ri.vision_layout(ids, cfg.video_token_id, cfg.image_token_id, grids, second_per_grid_ts,
                                               merge_size=cfg.vision.spatial_merge_size, tokens_per_second=cfg.vision.tokens_per_second)
        if embeds is not None and vis_pos.shape[0] != embeds.shape[0]:
            raise ValueError(f"Video features and video tokens do not match, tokens: {vis_pos.shape[0]}, features: {embeds.shape[0]}")
        if embeds is None and vis_pos.shape[0]:
            raise ValueError("prompt has video tokens but no video features were given")
        return pos, delta, vis_pos

    @_on_stream
    def prefill(self, W: Params, input_ids, video_embeds, plan, second_per_grid_ts, tag="pol", prefix_out=None, all_logits=None) -> PromptState:
        """input_ids: 1-D unpadded prompt (host ints).  video_embeds: window-ordered merged tokens from encode_video /
        encode_videos over the prompt's vision items in prompt order; second_per_grid_ts: one entry per item, None = still image.
        prefix_out = (k, v) tensors [L][>=S][nkv*hd] to write the prompt KV into (one slot of a multi-group batch)."""
        cfg, t = self.cfg, self.cfg.text
        ids = np.asarray(input_ids, dtype=np.int64).reshape(-1)
        S = int(ids.shape[0])
        pos, delta, vid_pos = self.vision_layout(ids, plan, second_per_grid_ts, video_embeds)
        H, hd, nkv = t.hidden_size, t.head_dim, t.num_key_value_heads
        d_ids = _i32(ids, self.device)
        x = self.buf(tag + ".px", (S, H), torch.float32)
        lib.call("vr2_embed_rows", d_ids, W["embed"], W["embed"].stride(0), x, x.stride(0), S, H, self.s)
        if video_embeds is not None:
            # row i of the window-ordered merger output is natural token window_index[i] (HF :464-466)
            dst = _i32(vid_pos[plan.window_index], self.device)
            lib.call("vr2_scatter_rows_f32", video_embeds, video_embeds.stride(0), dst, x, x.stride(0), vid_pos.shape[0], H, self.s)
        d_pos = _i32(pos, self.device)
        cos = self.buf(tag + ".pcos", (S, hd // 2), torch.float32)
        sin = self.buf(tag + ".psin", (S, hd // 2), torch.float32)
        lib.call("vr2_mrope_table", d_pos, S, self.inv_freq, self.section, cos, sin, hd, self.s)
        if prefix_out is not None:
            pk, pv = prefix_out
            assert pk.shape[1] >= S
        else:
            pk = self.buf(tag + ".prefix_k", (t.num_hidden_layers, S, nkv * hd), torch.bfloat16)
            pv = self.buf(tag + ".prefix_v", (t.num_hidden_layers, S, nkv * hd), torch.bfloat16)
        work = _i32(vp.causal_work(S, 0, 0, 0, 0, q_tile=vp.LM_Q_TILE), self.device)
        self._lm_rows(W, x, S, cos, sin, work, None, None, pk, pv, 0, tag + ".p")
        slot = "" if prefix_out is None else str(pk.data_ptr())
        last = self.buf(tag + ".plast" + slot, (1, H), torch.float32)
        last.copy_(x[S - 1:S])
        logits = self.buf(tag + ".plogits" + slot, (1, t.vocab_size), torch.float32)
        self._logits_rows(W, x[S - 1:S], 1, logits)
        if all_logits is not None:        # f32 [S, V] of the caller: the logits of EVERY position (HF-style forward().logits)
            for r0 in range(0, S, 1024):
                n = min(1024, S - r0)
                self._logits_rows(W, x[r0:r0 + n], n, all_logits[r0:r0 + n])
        return PromptState(n_prompt=S, rope_delta=delta, prefix_k=pk, prefix_v=pv, last_hidden=last, last_logits=logits, input_ids=ids)

    @_on_stream
    def prefill_batch(self, W: Params, prompts, slots, tag="pol", max_rows=36864):
        """Prompt prefill of several prompts in ONE row space (M = sum of rows: the decoder GEMMs run on full tiles instead of
        one ~1k-row prompt at a time).  prompts: list of (input_ids, video_embeds, plan, second_per_grid_ts) as for prefill();
        slots: Engine.prefix_slots(...) -- prompt p's K/V go to slots[p].  Rows per prompt = the longest prompt of the batch
        (shorter ones are padded with pad tokens after their last token: causal attention never looks ahead, the padded rows
        are never read).  -> [PromptState]; long batches are cut into pieces of <= max_rows rows."""
        cfg, t = self.cfg, self.cfg.text
        H, hd, nkv, L, V = t.hidden_size, t.head_dim, t.num_key_value_heads, t.num_hidden_layers, t.vocab_size
        idss = [np.asarray(p[0], dtype=np.int64).reshape(-1) for p in prompts]
        rp = max(len(i) for i in idss)
        per = max(1, max_rows // rp)
        if len(prompts) > per:
            out = []
            for c0 in range(0, len(prompts), per):
                out += self.prefill_batch(W, prompts[c0:c0 + per], slots[c0:c0 + per], tag=tag, max_rows=max_rows)
            return out
        P = len(prompts)
        k0, v0 = slots[0]
        kvw = nkv * hd
        if k0.stride(2) != 1 or k0.stride(1) != kvw or k0.shape[1] < rp:
            raise ValueError("prefill_batch needs cache slots from Engine.prefix_slots with room for the longest prompt")
        gstride = 0
        if P > 1:
            gstride = (slots[1][0].data_ptr() - k0.data_ptr()) // (k0.element_size() * kvw)       # slot stride in rows
            for p_, (sk, _) in enumerate(slots):
                if sk.data_ptr() != k0.data_ptr() + p_ * gstride * kvw * k0.element_size():
                    raise ValueError("prefill_batch needs cache slots with a uniform stride (Engine.prefix_slots)")
        lstride = k0.stride(0) // kvw                           # rows between layers inside a slot
        T = P * rp
        ids_all = np.full((P, rp), cfg.pad_token_id, dtype=np.int64)
        pos_all = np.zeros((3, P, rp), dtype=np.int64)
        deltas, works, vis_pos = [], [], []
        for p_, ((_, emb, plan, sec), ids) in enumerate(zip(prompts, idss)):
            S = len(ids)
            ids_all[p_, :S] = ids
            pos, delta, vis = self.vision_layout(ids, plan, sec, emb)
            pos_all[:, p_, :S] = pos
            deltas.append(delta)
            vis_pos.append(vis)
            # queries = this prompt's rows of the batch; keys = its own cache slot, causal (row index relative to slot 0 of the layer)
            works.append(vp.causal_work(S, p_ * rp, 0, 0, p_ * gstride, q_tile=vp.LM_Q_TILE))
        x = self.buf(tag + ".bx", (T, H), torch.float32)
        d_ids = _i32(ids_all.reshape(-1), self.device)
        lib.call("vr2_embed_rows", d_ids, W["embed"], W["embed"].stride(0), x, x.stride(0), T, H, self.s)
        for p_, ((_, emb, plan, _), ids) in enumerate(zip(prompts, idss)):
            if emb is not None:
                vid_pos = vis_pos[p_]
                dst = _i32(p_ * rp + vid_pos[plan.window_index], self.device)
                lib.call("vr2_scatter_rows_f32", emb, emb.stride(0), dst, x, x.stride(0), vid_pos.shape[0], H, self.s)
        cos = self.buf(tag + ".bcos", (T, hd // 2), torch.float32)
        sin = self.buf(tag + ".bsin", (T, hd // 2), torch.float32)
        lib.call("vr2_mrope_table", _i32(pos_all.reshape(3, T), self.device), T, self.inv_freq, self.section, cos, sin, hd, self.s)
        work = _i32(np.concatenate(works, 0), self.device)
        # per-layer K/V base = layer l of slot 0: [L][rows][kv] view with the slot's own layer stride
        kl = torch.as_strided(k0, (L, lstride, kvw), (lstride * kvw, kvw, 1))
        vl = torch.as_strided(v0, (L, lstride, kvw), (lstride * kvw, kvw, 1))
        self._lm_rows(W, x, T, cos, sin, work, None, None, kl, vl, 0, tag + ".b", seg=(rp, max(gstride, rp)))
        last_rows = torch.as_tensor([p_ * rp + len(ids) - 1 for p_, ids in enumerate(idss)], device=self.device)
        lasts = torch.index_select(x, 0, last_rows)             # owned by the returned states (a batch may be cut into pieces)
        logits = torch.empty((P, V), dtype=torch.float32, device=self.device)
        self._logits_rows(W, lasts, P, logits)
        return [PromptState(n_prompt=len(ids), rope_delta=deltas[p_], prefix_k=slots[p_][0], prefix_v=slots[p_][1],
                            last_hidden=lasts[p_:p_ + 1], last_logits=logits[p_:p_ + 1], input_ids=ids) for p_, ids in enumerate(idss)]

    # ------------------------------------------------------------------ decode (K13, K14)
    def prefix_slots(self, n_groups, max_prompt, tag="pol"):
        """One buffer holding the prompt KV of n_groups prompt groups with a uniform group stride: [(k, v)] per group."""
        t = self.cfg.text
        kv = t.num_key_value_heads * t.head_dim
        bk = self.buf(tag + ".slots_k", (n_groups, t.num_hidden_layers, max_prompt, kv), torch.bfloat16)
        bv = self.buf(tag + ".slots_v", (n_groups, t.num_hidden_layers, max_prompt, kv), torch.bfloat16)
        return [(bk[p], bv[p]) for p in range(n_groups)]

    def _decode_step(self, W: Params, st, G, C, bufs, temperature, top_k, seed_dev, eos_id, pad_id, top_p=1.0, min_p=0.0, rep=1.0):
        t = self.cfg.text
        H, hd, nq, nkv, I, V = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size, t.vocab_size
        nqkv = (nq + 2 * nkv) * hd
        x, xn, q, att, act = bufs["x"], bufs["xn"], bufs["q"], bufs["att"], bufs["act"]
        p_qkv, p_o, p_d, logits = bufs["p_qkv"], bufs["p_o"], bufs["p_d"], bufs["logits"]
        ns_q, ns_o, ns_d = p_qkv.shape[0], p_o.shape[0], p_d.shape[0]
        lib.call("vr2_embed_rows", bufs["next_ids"], W["embed"], W["embed"].stride(0), x, x.stride(0), G, H, self.s)
        scale = hd ** -0.5
        # config 5 above 16 rows: the norms also emit the e4m3 quantisation of their output (fused), att and act get their own pass
        fq = self._qbuf(G) if (self._a8w8(W, "l0.qkv_w", G) and H % 64 == 0 and H <= 4096 and H // 128 >= max(ns_q, 1)) else None
        for l in range(t.num_hidden_layers):
            if l == 0:
                self._norm(x, W["l0.ln1"], xn, G, H, t.rms_norm_eps, q8=fq)
            else:
                self._norm(x, W[f"l{l}.ln1"], xn, G, H, t.rms_norm_eps, partial=p_d, nsplit=ns_d, q8=fq)
            self._gemv(xn, W, f"l{l}.qkv_w", G, p_qkv, ns_q, prequant=fq)
            sk, sv = bufs["suffix_k"][l], bufs["suffix_v"][l]
            lib.call("vr2_rope_append_decode", p_qkv, p_qkv.stride(0), p_qkv.stride(1), ns_q, W[f"l{l}.qkv_b"], bufs["pos"],
                     bufs["step"], self.inv_freq, G, nq, nkv, hd, q, q.stride(0), sk, sv, sk.stride(0), C, self.s)
            lib.call("vr2_attn_decode", q, q.stride(0), bufs["pk"][l], bufs["pv"][l], sk, sv, sk.stride(0), bufs["plen"],
                     bufs["spg"], bufs["pstride"], C, bufs["step"], bufs["schunk"], bufs["pchunk"], bufs["npc"], bufs["nsc"], G, nq, t.group, hd, scale,
                     bufs["part_o"], bufs["part_ml"], att, att.stride(0), bufs.get("skip_finished"), self.s)
            self._gemv(att, W, f"l{l}.o_w", G, p_o, ns_o)
            self._norm(x, W[f"l{l}.ln2"], xn, G, H, t.rms_norm_eps, partial=p_o, nsplit=ns_o, q8=fq)
            f8 = self._f8(W, f"l{l}.gu_w", G)
            if f8 is not None and self._a8w8(W, f"l{l}.gu_w", G):
                q8, xs = fq if fq is not None else self._quant(xn, G, H)
                lib.call("vr2_gemv_gateup_silu_a8w8", q8, q8.stride(0), xs, G, f8[0], f8[1], I, H, act, act.stride(0), self.s)
            elif f8 is not None:
                lib.call("vr2_gemv_gateup_silu_w8", xn, xn.stride(0), G, f8[0], f8[1], I, H, act, act.stride(0), self.s)
            else:
                lib.call("vr2_gemv_gateup_silu_bf16", xn, xn.stride(0), G, W[f"l{l}.gu_w"], I, H, act, act.stride(0), self.s)
            self._gemv(act, W, f"l{l}.down_w", G, p_d, ns_d)
        self._norm(x, W["norm"], xn, G, H, t.rms_norm_eps, partial=p_d, nsplit=ns_d, q8=fq)
        self._gemv(xn, W, "lm_head", G, logits, 1, prequant=fq)
        # token index step+1 is sampled from the logits of token `step`
        tok1 = bufs["tokens"][:, 1:]
        if rep != 1.0:    # HF's processors run before its warpers; next_ids still holds the token this step consumed: mark it seen
            lib.call("vr2_repetition_penalty", logits, logits.stride(0), G, V, rep, bufs["seen"], bufs["seen"].stride(0), bufs["next_ids"], self.s)
        lib.call("vr2_sample_topk_p", logits, logits.stride(0), G, V, temperature, top_k, top_p, min_p, 0, seed_dev, bufs["step"],
                 bufs["finished"], eos_id, pad_id, bufs["next_ids"], tok1, bufs["tokens"].stride(0), bufs["cand"], self.s)
        lib.call("vr2_add_int", bufs["step"], 1, self.s)

    def _prefix_chunk(self, n_groups, n_prompt):
        """Prompt keys per shared-prefix attention workgroup: the shortest multiple of 64 that covers the prompt with at most ONE
        workgroup per CU (groups x KV heads x chunks <= 256), between 64 and 512.  The workgroups are latency-bound walks over
        their 64-key tiles, so a second round on some CUs costs more than slightly longer walks on all of them.  Measured at
        16 groups, S = 1088 (the launch at a 32-token suffix, profiles/r02_kernel_stats_default_bench_p16.md): 17.8 / 18.7 /
        17.0 / 23.0 us for 192 / 256 / 320 / 576; 22.7 / 19.1 for 64 / 128 in round 1."""
        if PREFIX_CHUNK:
            return PREFIX_CHUNK
        nkv = self.cfg.text.num_key_value_heads
        per = max(1, 256 // max(1, n_groups * nkv))               # chunks one (group, KV head) may have
        pc = -(-(-(-n_prompt // per)) // 64) * 64                  # ceil(ceil(S / per) / 64) * 64
        return min(512, max(64, pc))

    def _suffix_chunk(self, n_seqs, n_new):
        """Own-suffix keys per decode-attention wave (four waves per workgroup): the shortest multiple of 64 that covers the
        n_new generated tokens with at most one suffix workgroup per CU (sequences x KV heads x chunks <= 1024 waves), between 64
        and 1024 -- the same rule as _prefix_chunk.  Measured at 128 sequences, C = 1024, the whole launch per layer: 51.2 /
        42.8 / 42.1 / 40.6 / 43.7 / 46.1 us for 128 / 256 / 384 / 512 / 768 / 1024 keys; at 64 sequences 256 was the round-1
        optimum, at 8 sequences 64."""
        if DECODE_CHUNK:
            return DECODE_CHUNK
        per = max(1, 1024 // max(1, n_seqs * self.cfg.text.num_key_value_heads))   # chunks one (sequence, KV head) may have
        sc = -(-(-(-n_new // per)) // 64) * 64
        return min(1024, max(64, sc))

    def _decode_bufs(self, G, C, n_prompt, pchunk=64):
        t = self.cfg.text
        H, hd, nq, nkv, I, V, L = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size, t.vocab_size, t.num_hidden_layers
        schunk = self._suffix_chunk(G, C)
        npc, nsc = (n_prompt + pchunk - 1) // pchunk, (C + schunk - 1) // schunk
        nchunk = npc + nsc
        # grid-level split-K per projection (profiles/src/tune_gemv.py on MI355X): enough workgroups to cover 256 CUs
        def pick(k, want):
            while want > 1 and (k // 32) < want * 4:
                want //= 2
            while G > 128 and want > 1 and (k // 64) < want * 4:     # above 128 rows only the ring kernel runs: >= 4 whole 64-k chunks per part
                want //= 2
            return want
        # (M <= 16: 32-row workgroups with 4 k-splitting waves; M > 16: the LDS-staged variant covers 128 output rows per
        # workgroup with 2 k groups -> the same slab count for q/k/v and o, twice for down)
        wq, wd = (4, 4) if G <= 16 else (4, 8)     # ring kernel at 128 rows, (q/o parts, down parts): (7,8) 4.84, (4,8) 4.72-4.77,
                                                    # (4,4) 5.03, (2,4) 5.20, (4,10) 4.70, (5,8) 4.73 ms/token (C = 256)
        if os.environ.get("VR2_SPLITS"):                               # measurement override: "q,d"
            wq, wd = (int(v) for v in os.environ["VR2_SPLITS"].split(","))
        if self.decode_fp8 and G <= (256 if self.decode_fp8_act else self.decode_fp8_max_rows):
            # fp8 weights: 64-k double steps (direct kernel) / 256-k chunks (staged kernel, no second k group): fewer, longer parts
            def pick(k, want):                     # noqa: F811
                while want > 1 and (k // 64) < want * 4:
                    want -= 1
                return want
            wq, wd = (4, 4) if G <= 16 else (7, 16) if G <= 64 else (7, 8)
        ns_q, ns_o, ns_d = pick(H, wq), pick(nq * hd, wq), pick(I, wd)
        b = dict(
            x=self.buf("d.x", (G, H), torch.float32), xn=self.buf("d.xn", (G, H), torch.bfloat16),
            q=self.buf("d.q", (G, nq * hd), torch.bfloat16), att=self.buf("d.att", (G, nq * hd), torch.bfloat16),
            act=self.buf("d.act", (G, I), torch.bfloat16),
            p_qkv=self.buf("d.p_qkv", (ns_q, G, (nq + 2 * nkv) * hd), torch.float32),
            p_o=self.buf("d.p_o", (ns_o, G, H), torch.float32), p_d=self.buf("d.p_d", (ns_d, G, H), torch.float32),
            logits=self.buf("d.logits", (G, V), torch.float32),
            suffix_k=self.buf("d.suffix_k", (L, G * C, nkv * hd), torch.bfloat16),
            suffix_v=self.buf("d.suffix_v", (L, G * C, nkv * hd), torch.bfloat16),
            part_o=self.buf("d.part_o", (G * nq * nchunk, hd), torch.float32),
            part_ml=self.buf("d.part_ml", (G * nq * nchunk, 2), torch.float32),
            tokens=self.buf("d.tokens", (G, C + 1), torch.int32), next_ids=self.buf("d.next", (G,), torch.int32),
            finished=self.buf("d.fin", (G,), torch.int32), step=self.buf("d.step", (1,), torch.int32),
            zero=self.buf("d.zero", (1,), torch.int32, zero=True), pos=self.buf("d.pos", (G,), torch.int32),
            npc=npc, nsc=nsc, pchunk=pchunk, schunk=schunk, cand=self.buf("d.cand", (int(lib.load().vr2_sample_ws_bytes(G)) // 4,), torch.float32),
        )
        return b

    def max_decode_rows(self):
        """sequences per decode weight pass: 256 on bf16 weights (round 4: the register-ring GEMV with 12 / 16 row blocks per wave;
        at 256 rows a token costs ~1.5x a 128-row token, so two groups' worth of sequences share one stream of the weights), 128 on
        weight-only fp8 copies (that kernel stops there; fp8 activations x fp8 weights run the same ring structure up to 256)"""
        return 128 if (self.decode_fp8 and not self.decode_fp8_act) else 256

    @_on_stream
    def decode(self, W: Params, st, G, max_new_tokens, temperature=0.9, top_k=50, seed=0, eos_token_id=None,
               use_graph=True, forced_tokens=None, check_every=64, top_p=1.0, min_p=None, repetition_penalty=1.0, finish_at=None):
        """Sample G completions per prompt group of up to max_new_tokens.  ``st``: one PromptState, or a list of P states whose
        prefix KV live in ``prefix_slots`` (P prompt groups decoded together: P*G <= max_decode_rows() rows per weight pass, so the weights
        are streamed once for all of them).  -> (int32 [P*G, C'] device, buffers).
        eos_token_id=None disables EOS (fixed length).  forced_tokens [P*G,C] (host ints) teacher-forces the inputs
        (the sampled tokens are still produced, from the same logits, for comparison).
        finish_at (host ints [P*G], benchmarks only): EOS positions EMULATED for models that never emit one (random-init
        weights): at every check point the sequences that have produced finish_at[g] tokens are flagged finished, exactly as if
        they had sampled their EOS there -- the decode stops once all are, finished sequences stop walking their own keys.
        top_p / min_p / repetition_penalty: HF generate's warpers and processor (grpo_trainer.py:559-570 forwards them); top_p < 1
        and min_p > 0 need 1 <= top_k <= 64; the penalty covers prompt + generated ids (PromptState.input_ids).

        One decode step (all layers + lm_head + sampling) is launch-bound (~260 kernels), so it is captured once per shape
        into a hipGraph; step index, next ids, finished flags, positions and the RNG seed live in device memory so the same
        graph is replayed for every token and every prompt."""
        import ctypes
        cfg = self.cfg
        states = st if isinstance(st, (list, tuple)) else [st]
        P = len(states)
        NG = P * G
        if NG > self.max_decode_rows():
            raise ValueError(f"{P} groups x {G} sequences = {NG} rows; the decode GEMV handles at most {self.max_decode_rows()} rows per pass"
                             + (" with fp8 decode weights" if self.decode_fp8 else ""))
        C = int(max_new_tokens)
        k0, v0 = states[0].prefix_k, states[0].prefix_v
        cap_s = k0.stride(0) // k0.stride(1)             # row capacity of the prefix cache: the graph is shaped for it
        b = self._decode_bufs(NG, C, cap_s, self._prefix_chunk(P, cap_s))
        V = cfg.text.vocab_size
        eos = -1 if eos_token_id is None else int(eos_token_id)
        seed_dev = self.buf("d.seed", (1,), torch.int64)
        forced = None
        if forced_tokens is not None:
            forced = torch.as_tensor(np.asarray(forced_tokens), dtype=torch.int32).to(self.device)
        # prefix KV addressing: group p at base + p * stride
        pstride = 0
        if P > 1:
            pstride = (states[1].prefix_k.data_ptr() - k0.data_ptr()) // k0.element_size()
            for p_, s_ in enumerate(states):
                if s_.prefix_k.data_ptr() != k0.data_ptr() + p_ * pstride * k0.element_size() or s_.prefix_k.stride() != k0.stride():
                    raise ValueError("multi-group decode needs prefix caches from Engine.prefix_slots (uniform group stride)")
        b["pk"], b["pv"], b["spg"], b["pstride"] = k0, v0, G, pstride
        # finished sequences (EOS sampled: pad tokens from then on) stop walking their own keys; not when the inputs are forced --
        # a teacher-forced row goes on whatever the sampler drew -- and not with EOS off
        b["skip_finished"] = b["finished"] if (forced_tokens is None and (eos >= 0 or finish_at is not None)) else None
        d_finish = None if finish_at is None else torch.as_tensor(np.asarray(finish_at), dtype=torch.int32).to(self.device)
        b["plen"] = self.buf("d.plen", (P,), torch.int32)
        pos0 = torch.tensor([s_.n_prompt + s_.rope_delta for s_ in states for _ in range(G)], dtype=torch.int32)
        plen = torch.tensor([s_.n_prompt for s_ in states], dtype=torch.int32)
        logits0 = self.buf("d.logits0", (NG, V), torch.float32)

        def reset():
            b["step"].zero_(); b["finished"].zero_(); b["tokens"].fill_(cfg.pad_token_id)
            b["pos"].copy_(pos0.to(self.device)); b["plen"].copy_(plen.to(self.device))
            seed_dev.fill_(int(seed) & (2**63 - 1))
            if seen0 is not None:
                b["seen"].copy_(seen0.to(self.device))

        top_p = 1.0 if top_p is None else float(top_p)
        min_p = 0.0 if min_p is None else float(min_p)
        rep = 1.0 if repetition_penalty is None else float(repetition_penalty)
        if (top_p < 1.0 or min_p > 0.0) and not 1 <= int(top_k) <= 64:
            raise ValueError(f"top_p={top_p} / min_p={min_p} run on the top-k candidate set: they need 1 <= top_k <= 64 (got {top_k})")
        seen0 = None
        if rep != 1.0:
            words = (V + 31) // 32
            b["seen"] = self.buf("d.seen", (NG, words), torch.int32)
            bits = np.zeros((NG, words * 32), dtype=bool)
            for p_, s_ in enumerate(states):
                if s_.input_ids is None:
                    raise ValueError("repetition_penalty needs PromptState.input_ids (states from Engine.prefill / prefill_batch)")
                ids_p = np.asarray(s_.input_ids).reshape(-1)
                bits[p_ * G:(p_ + 1) * G, ids_p[(ids_p >= 0) & (ids_p < V)]] = True
            seen0 = torch.from_numpy(np.packbits(bits, axis=1, bitorder="little").view(np.int32).copy())
        step_args = (W, None, NG, C, b, temperature, top_k, seed_dev, eos, cfg.pad_token_id, top_p, min_p, rep)
        key = (id(W), NG, P, C, cap_s, float(temperature), int(top_k), eos, k0.data_ptr(), pstride, self._ws_epoch, bool(self.decode_fp8),
               bool(self.decode_fp8_act), top_p, min_p, rep, b["skip_finished"] is not None)
        graph = None
        if getattr(self, "_graph_epoch", -1) != self._ws_epoch:       # a workspace moved: every captured graph is stale
            self._drop_graphs()
        if use_graph and forced is None and C > 1:
            graph = self._graphs.get(key)
            if graph is None:
                reset()
                b["next_ids"].zero_()
                self._decode_step(*step_args)          # eager warm-up: loads every code object before capture
                self.sync()
                lib.call("vr2_graph_begin", self.s)
                ex = ctypes.c_void_p()
                try:
                    self._decode_step(*step_args)
                finally:
                    rc = lib.load().vr2_graph_end(ctypes.c_void_p(self.s), ctypes.byref(ex))
                if rc != 0 or not ex.value:
                    raise lib.Vr2Error("graph capture failed: " + lib.load().vr2_last_error().decode())
                graph = ex
                self._graphs[key] = graph
        reset()
        # token 0: G draws per group from its prompt's last-position logits; its own RNG stream (seed arg)
        for p_, s_ in enumerate(states):
            logits0[p_ * G:(p_ + 1) * G].copy_(s_.last_logits.expand(G, V))
        tok0 = b["tokens"]
        if rep != 1.0:
            lib.call("vr2_repetition_penalty", logits0, logits0.stride(0), NG, V, rep, b["seen"], b["seen"].stride(0), None, self.s)
        lib.call("vr2_sample_topk_p", logits0, logits0.stride(0), NG, V, temperature, top_k, top_p, min_p, (int(seed) * 2654435761 + 12345) & (2**64 - 1),
                 None, b["zero"], b["finished"], eos, cfg.pad_token_id, b["next_ids"], tok0, tok0.stride(0), b["cand"], self.s)
        steps_done = 0
        for step in range(C - 1):
            if forced is not None:
                b["next_ids"].copy_(forced[:, step])
            if graph is not None:
                lib.call("vr2_graph_launch", graph, self.s)
            else:
                self._decode_step(*step_args)
            steps_done = step + 1
            if (eos >= 0 or d_finish is not None) and forced is None and (step + 1) % check_every == 0:
                if d_finish is not None:
                    b["finished"].copy_((d_finish <= step + 2).to(torch.int32))      # step + 2 tokens exist per sequence by now
                self.sync()
                if bool(b["finished"].all().item()):
                    break
        n_tok = steps_done + 1
        return b["tokens"][:, :n_tok].clone(), b     # clone: the workspace is reused by the next call

    # ------------------------------------------------------------------ teacher-forced log-probs (K15)
    @staticmethod
    def ragged_layout(G, C, lens):
        """Completions cut at their own lengths, packed back to back: lens[g] <= C rows for sequence g.  -> dict(lens, offs (row
        of each sequence's first token in the packed space), R (total rows), sel (int64 [R]: packed row -> flat index g*C + j of
        the padded [G, C] layout), first (int64: packed rows with j == 0), src (int64 [R]: packed row whose hidden state PREDICTS
        this row's token, i.e. row - 1; -1 for j == 0: the prompt's last row))."""
        lens = np.asarray(lens, dtype=np.int64).reshape(-1)
        if lens.shape[0] != G or (lens < 0).any() or (lens > C).any():
            raise ValueError(f"lens must hold one length in [0, {C}] per sequence, got {lens.tolist()}")
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        R = int(lens.sum())
        sel = np.concatenate([g * C + np.arange(lens[g]) for g in range(G)]) if R else np.zeros(0, dtype=np.int64)
        src = np.arange(R, dtype=np.int64) - 1
        first = offs[lens > 0]
        src[first] = -1
        return dict(lens=lens, offs=offs, R=R, sel=sel.astype(np.int64), first=first.astype(np.int64), src=src)

    HEAD_CHUNK = 4096     # rows per lm_head launch of the unified passes: 4096 x 152064 fp32 = 2.5 GB, full 256-row tiles

    def unified_layout(self, prompt_ids, plan, second_per_grid_ts, video_embeds, completion_ids, lens=None):
        """The ONE row space of a prompt group's teacher-forced pass: T = S + R rows = the S prompt rows (causal among themselves)
        followed by the completion rows of each of the G sequences (each sees the whole prompt + its own causal prefix), the
        sequences cut at lens[g] and packed back to back when lens is given.  Both passes whose difference is the loss -- the
        no-grad scoring (score_unified: reference model, old policy) and the training forward (train.Trainer.forward) -- are laid
        out by THIS function and run the same launches over it, so that identical weights give identical bits at every shape."""
        ids = np.asarray(prompt_ids, dtype=np.int64).reshape(-1)
        S = int(ids.shape[0])
        comp = torch.as_tensor(np.asarray(completion_ids) if not torch.is_tensor(completion_ids) else completion_ids)
        comp = comp.to(device=self.device, dtype=torch.int32).contiguous()
        G, C = comp.shape
        if lens is not None and int(np.asarray(lens).sum()) == G * C:
            lens = None                                           # nothing to cut: the dense layout
        lay = None if lens is None else self.ragged_layout(G, C, lens)
        R = G * C if lay is None else lay["R"]
        seq_len = [C] * G if lay is None else [int(n) for n in lay["lens"]]
        seq_off = [g * C for g in range(G)] if lay is None else [int(o) for o in lay["offs"]]    # first row of a sequence, from S
        pos, delta, vid_pos = self.vision_layout(ids, plan, second_per_grid_ts, video_embeds)
        pc = np.concatenate([np.arange(n, dtype=np.int64) + S + delta for n in seq_len]) if R else np.zeros(0, dtype=np.int64)
        pos_all = np.concatenate([pos, np.broadcast_to(pc[None, :], (3, R))], 1)
        d_sel = None if lay is None else torch.from_numpy(lay["sel"]).to(self.device)
        tok = comp.view(-1) if lay is None else comp.view(-1).index_select(0, d_sel).contiguous()
        work = vp.heaviest_first(np.concatenate([vp.causal_work(S, 0, 0, 0, 0, q_tile=vp.LM_Q_TILE)] +
                                                [vp.causal_work(n, S + o, 0, S, S + o, q_tile=vp.LM_Q_TILE) for n, o in zip(seq_len, seq_off) if n > 0], 0))
        # head row r is predicted by x row S + r - 1, a sequence's first row by the prompt's last row
        if lay is None:
            src = np.arange(R, dtype=np.int64) - 1
            src[np.arange(G) * C] = -1
            first = np.arange(G, dtype=np.int64) * C
        else:
            src, first = lay["src"], lay["first"]
        pred = torch.from_numpy(np.where(src < 0, S - 1, S + src)).to(self.device)
        return dict(ids=ids, S=S, G=G, C=C, R=R, T=S + R, lay=lay, seq_len=seq_len, seq_off=seq_off, pos_all=pos_all, vid_pos=vid_pos,
                    sel=d_sel, tok=tok, all_ids=torch.cat([_i32(ids, self.device), tok]), work=_i32(work, self.device), pred=pred,
                    first=torch.from_numpy(np.asarray(first, dtype=np.int64)).to(self.device))

    def unified_inputs(self, W: Params, u, video_embeds, plan, tag):
        """embedding rows (+ the scattered vision features) and the rotary tables of a unified row space -> (x f32 [T][H], cos, sin)"""
        t = self.cfg.text
        T, H, hd = u["T"], t.hidden_size, t.head_dim
        x = self.buf(tag + ".x", (T, H), torch.float32)
        lib.call("vr2_embed_rows", u["all_ids"], W["embed"], W["embed"].stride(0), x, x.stride(0), T, H, self.s)
        if video_embeds is not None:
            dst = _i32(u["vid_pos"][plan.window_index], self.device)
            lib.call("vr2_scatter_rows_f32", video_embeds, video_embeds.stride(0), dst, x, x.stride(0), u["vid_pos"].shape[0], H, self.s)
        cos = self.buf(tag + ".cos", (T, hd // 2), torch.float32)
        sin = self.buf(tag + ".sin", (T, hd // 2), torch.float32)
        lib.call("vr2_mrope_table", _i32(u["pos_all"], self.device), T, self.inv_freq, self.section, cos, sin, hd, self.s)
        return x, cos, sin

    def head_logps(self, W: Params, hs, u, temperature, logp, hn=None, logits_all=None, lse_out=None, allow_f8=True):
        """log-prob head over the R gathered rows hs (fp32): final norm, lm_head in HEAD_CHUNK-row launches, log-softmax at the
        given token -> logp[R].  hn / logits_all / lse_out: buffers of the training forward that keep the normed rows, the logits
        and the log-sum-exps for its backward."""
        V = self.cfg.text.vocab_size
        R = u["R"]
        rc = min(self.HEAD_CHUNK, R)
        lg = None if logits_all is not None else self.buf("lm.chunk_logits", (rc, V), torch.float32)
        for r0 in range(0, R, rc):
            n = min(rc, R - r0)
            if logits_all is not None:
                lg = logits_all[r0:r0 + n]
            self._logits_rows(W, hs[r0:r0 + n], n, lg, xn=None if hn is None else hn[r0:r0 + n], allow_f8=allow_f8)
            lib.call("vr2_lse_gather", lg, lg.stride(0), n, V, u["tok"][r0:r0 + n], temperature, logp[r0:r0 + n],
                     None if lse_out is None else lse_out[r0:r0 + n], self.s)

    def unified_logps_out(self, u, logp):
        """packed log-probs [R] -> f32 [G, C] (0 at the positions behind a sequence's length)"""
        if u["lay"] is None:
            return logp.view(u["G"], u["C"])
        out = torch.zeros(u["G"] * u["C"], dtype=torch.float32, device=self.device)
        out.index_copy_(0, u["sel"], logp)
        return out.view(u["G"], u["C"])

    @_on_stream
    def score_unified(self, W: Params, prompt_ids, video_embeds, plan, second_per_grid_ts, completion_ids, temperature, tag="ref", lens=None,
                      allow_fp8=True):
        """Per-token log-probs f32 [G, C] of ``_get_per_token_logps`` (grpo_trainer.py:723-741) for one prompt group, computed over
        the unified [prompt | completions] row space -- the SAME row space, work list and launches as the policy's training forward
        (train.Trainer.forward), without saved activations.  This is the pass for the log-probs that enter the loss beside the
        policy's: the frozen reference model (:826) and the old policy (:814-821).  With identical weights the two passes give
        identical bits at every shape (so kl == 0 and ratio == 1 exactly where the reference has them exactly), which the
        prefix-cache pass (``score``: fewer rows, other tilings, other k orders in the tile tails) does not.
        Cost: the S prompt rows are recomputed here instead of read from a prefix cache -- the row count a prompt prefill plus a
        cached scoring pass have together.
        allow_fp8=False keeps this call on the bf16 GEMMs although the engine's gemm_fp8 mode (config 5) is on: the old-policy
        log-probs must come from the arithmetic of the training forward (bf16), or the first-iteration ratio is not 1."""
        t = self.cfg.text
        H, hd, nkv = t.hidden_size, t.head_dim, t.num_key_value_heads
        u = self.unified_layout(prompt_ids, plan, second_per_grid_ts, video_embeds, completion_ids, lens)
        if u["R"] == 0:
            return torch.zeros((u["G"], u["C"]), dtype=torch.float32, device=self.device)
        T, R = u["T"], u["R"]
        mode = self.gemm_fp8
        self.gemm_fp8 = mode and allow_fp8
        try:
            x, cos, sin = self.unified_inputs(W, u, video_embeds, plan, tag + ".u")
            ks = self.buf(tag + ".uk", (T, nkv * hd), torch.bfloat16)
            vs = self.buf(tag + ".uv", (T, nkv * hd), torch.bfloat16)
            self._lm_rows(W, x, T, cos, sin, u["work"], None, None, ks, vs, 0, tag + ".u")
            hs = self.buf(tag + ".uhs", (R, H), torch.float32)
            torch.index_select(x, 0, u["pred"], out=hs)
            logp = self.buf(tag + ".ulogp", (R,), torch.float32)
            self.head_logps(W, hs, u, temperature, logp)
        finally:
            self.gemm_fp8 = mode
        return self.unified_logps_out(u, logp)

    @_on_stream
    def score(self, W: Params, st: PromptState, completion_ids, temperature, tag="pol", row_chunk=1024, return_logits=False, lens=None):
        """completion_ids int [G,C] -> per-token log-probs f32 [G,C] (device) of ``_get_per_token_logps``.

        Row (g,j) of the lm_head input is the hidden state that PREDICTS completion token j: the prompt's last token for
        j=0 (shared by all g), completion token j-1 otherwise.  Completion rows attend the shared prompt prefix KV plus
        their own causal prefix.
        lens (host ints, one per sequence): only the first lens[g] tokens of sequence g are scored -- the rows behind a
        completion's first EOS are masked out of the loss (grpo_trainer.py:793-806), so they are not computed at all: the
        sequences are packed back to back (sum(lens) rows instead of G*C) and the skipped positions read 0."""
        cfg, t = self.cfg, self.cfg.text
        comp = torch.as_tensor(np.asarray(completion_ids) if not torch.is_tensor(completion_ids) else completion_ids)
        comp = comp.to(device=self.device, dtype=torch.int32).contiguous()
        G, C = comp.shape
        H, hd, nkv, V = t.hidden_size, t.head_dim, t.num_key_value_heads, t.vocab_size
        S = st.n_prompt
        if lens is not None and int(np.asarray(lens).sum()) == G * C:
            lens = None                                           # nothing to cut: the dense layout
        if lens is None:
            T = G * C
            flat_tok = comp.view(-1)
            p1 = (np.arange(C, dtype=np.int64) + S + st.rope_delta)
            pos1 = np.tile(p1, G)
            works = [vp.causal_work(C, g * C, 0, S, g * C, q_tile=vp.LM_Q_TILE) for g in range(G)]
            lay = None
        else:
            if return_logits:
                raise ValueError("return_logits needs the dense layout (lens=None)")
            lay = self.ragged_layout(G, C, lens)
            T = lay["R"]
            if T == 0:
                return torch.zeros((G, C), dtype=torch.float32, device=self.device)
            d_sel = torch.from_numpy(lay["sel"]).to(self.device)
            flat_tok = comp.view(-1).index_select(0, d_sel).contiguous()
            pos1 = np.concatenate([np.arange(n, dtype=np.int64) + S + st.rope_delta for n in lay["lens"]])
            works = [vp.causal_work(int(n), int(o), 0, S, int(o), q_tile=vp.LM_Q_TILE) for n, o in zip(lay["lens"], lay["offs"]) if n > 0]
        x = self.buf(tag + ".sx", (T, H), torch.float32)
        lib.call("vr2_embed_rows", flat_tok, W["embed"], W["embed"].stride(0), x, x.stride(0), T, H, self.s)
        # positions of completion token j: S + j + rope_delta on all three axes
        pos = np.broadcast_to(pos1[None, :], (3, T))
        d_pos = _i32(pos, self.device)
        cos = self.buf(tag + ".scos", (T, hd // 2), torch.float32)
        sin = self.buf(tag + ".ssin", (T, hd // 2), torch.float32)
        lib.call("vr2_mrope_table", d_pos, T, self.inv_freq, self.section, cos, sin, hd, self.s)
        ks = self.buf(tag + ".sk", (T, nkv * hd), torch.bfloat16)
        vs = self.buf(tag + ".sv", (T, nkv * hd), torch.bfloat16)
        work = vp.heaviest_first(np.concatenate(works, 0))
        d_work = _i32(work, self.device)
        self._lm_rows(W, x, T, cos, sin, d_work, st.prefix_k, st.prefix_v, ks, vs, 0, tag + ".s")
        # shift: hs[(g,0)] = prompt last hidden ; hs[(g,j)] = x[(g,j-1)]
        hs = self.buf(tag + ".shs", (T, H), torch.float32)
        with torch.cuda.stream(self.stream):
            if lay is None:
                hv, xv = hs.view(G, C, H), x.view(G, C, H)
                hv[:, 1:].copy_(xv[:, :-1])
                hv[:, 0].copy_(st.last_hidden.expand(G, H))
            else:
                torch.index_select(x, 0, torch.from_numpy(np.maximum(lay["src"], 0)).to(self.device), out=hs)
                hs.index_copy_(0, torch.from_numpy(lay["first"]).to(self.device), st.last_hidden.expand(len(lay["first"]), H))
        logp = self.buf(tag + ".slogp", (T,), torch.float32)
        rc = min(row_chunk, T)
        lg = self.buf("lm.chunk_logits", (rc, V), torch.float32)
        keep = [] if return_logits else None
        for r0 in range(0, T, rc):
            n = min(rc, T - r0)
            self._logits_rows(W, hs[r0:r0 + n], n, lg)
            lib.call("vr2_lse_gather", lg, lg.stride(0), n, V, flat_tok[r0:r0 + n], temperature, logp[r0:r0 + n], None, self.s)
            if return_logits:
                with torch.cuda.stream(self.stream):
                    keep.append(lg[:n].clone())
        if lay is not None:
            with torch.cuda.stream(self.stream):
                out = torch.zeros(G * C, dtype=torch.float32, device=self.device)
                out.index_copy_(0, d_sel, logp)
            return out.view(G, C)
        out = logp.view(G, C)
        if return_logits:
            return out, torch.cat(keep, 0).view(G, C, V)
        return out
