"""Training the vision side: `--freeze_vision_tower False` and / or `--freeze_merger False`.

The reference sets requires_grad on `model.visual` from `freeze_vision_tower` and then on `model.visual.merger` from
`freeze_merger` (/root/reference/train/src/train/train_grpo.py:48-57, called at :123-124); the GRPO trainer optimises whatever
requires grad with ONE learning rate (vision_lr / merger_lr are read by the SFT trainer only).  The shipped recipe freezes both
(train_grpo.sh:47-49), which is the engine's default path and is untouched by this module.

When either trains, the policy's training forward (train.Trainer.forward) runs the vision tower through `VisionTrainer.forward`
instead of reusing the step's no-grad embeddings: the same launches as Engine._vision_tower / _vision_merger (bit-identical
embeddings) with every layer's inputs kept, and `VisionTrainer.backward` takes d loss / d (video token rows) -- the LM backward's
input gradient at the scattered rows -- down through the merger (fc2, erf-GELU, fc1, ln_q) and, if the tower trains, the ViT blocks
(SwiGLU MLP with biases, RMSNorm, rotary, non-causal windowed / per-frame attention, patch embedding).  Every piece is an existing
kernel of the LM backward at the ViT's widths; the 80-wide heads go through the head_dim-128 attention backward zero-padded (the
padding columns contribute nothing to scores, outputs or gradients), with the forward's own log-sum-exps.

Mixed precision as for the LM (train.py): bf16 GEMM operands, fp32 accumulation / residual / norm gradients, fp32 masters and Adam
moments, the AdamW kernel re-emits the tile-packed bf16 copies (W for the forward, W^T for the dgrad GEMMs) in place -- the next
rollout's Engine.encode_videos reads the updated tower.
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib
from .engine import EPI_BF16, EPI_F32, EPI_ADD_F32, Engine, _i32
from .host import vision_plan as vp
from .weights import Params, pack_tiles, pad64, unpack_tiles


class VisionState:
    """fp32 masters, gradients (one contiguous arena = one data-parallel bucket, "vision"), Adam moments and the transposed
    packed copies of the trainable vision tensors."""

    def __init__(self, vision: Params, device, train_tower, train_merger):
        self.params = vision
        self.device = device
        self.master, self.grad, self.m, self.v, self.kind, self.decay = {}, {}, {}, {}, {}, {}
        v = vision.cfg.vision
        mats, vecs = [], []
        if train_merger:
            mats += ["fc1_w", "fc2_w"]
            vecs += ["ln_q", "fc1_b", "fc2_b"]
        if train_tower:
            mats += ["patch_w"] + [f"b{i}.{n}" for i in range(v.depth) for n in ("qkv_w", "proj_w", "gu_w", "down_w")]
            vecs += [f"b{i}.{n}" for i in range(v.depth) for n in ("norm1", "norm2", "qkv_b", "proj_b", "gu_b", "down_b")]
        through = ["fc1_w", "fc2_w"] if train_tower and not train_merger else []   # frozen merger above a trainable tower: the
        for name in mats + through:                                                # gradient still passes through its W^T
            n, kp = vision.shape[name]
            w = unpack_tiles(vision[name], n, kp)                    # [N][Kp] bf16 (K padding columns are zero)
            if name not in through:
                self._add(name, w.float(), "mat", True)
            wt = torch.zeros((kp, pad64(n)), dtype=torch.bfloat16, device=device)
            wt[:, :n] = w.t()
            vision.t[name + "_t"] = pack_tiles(wt)
            vision.shape[name + "_t"] = (kp, pad64(n))
        for name in vecs:
            self._add(name, vision[name], "vec", False)              # the fp32 vector IS its own master (updated in place)
        total = sum(p.numel() for p in self.master.values())
        self.arena = torch.zeros(total, dtype=torch.float32, device=device)
        at = 0
        for name, p in self.master.items():
            self.grad[name] = self.arena[at:at + p.numel()].view(p.shape)
            at += p.numel()
        self.buckets = [("vision", 0, total)]

    def _add(self, name, master, kind, decay):
        self.master[name] = master.contiguous()
        self.m[name] = torch.zeros_like(self.master[name])
        self.v[name] = torch.zeros_like(self.master[name])
        self.kind[name] = kind
        self.decay[name] = decay

    def zero_grad(self):
        self.arena.zero_()

    def n_params(self):
        return sum(p.numel() for p in self.master.values())


class VisionTrainer:
    def __init__(self, engine: Engine, vision: Params, train_tower: bool, train_merger: bool):
        if not (train_tower or train_merger):
            raise ValueError("VisionTrainer: nothing to train")
        self.eng = engine
        self.W = vision
        self.train_tower, self.train_merger = bool(train_tower), bool(train_merger)
        self.state = VisionState(vision, engine.device, train_tower, train_merger)
        self._saved = None

    # ------------------------------------------------------------------ helpers
    def _gemm(self, A, wname, bias, C, M, epi):
        self.eng._gemm(A, self.W, wname, bias, C, M, epi)

    def _wgrad(self, dY, X, name, M_out, N_in, T):
        g = self.state.grad[name]
        lib.call("vr2_gemm_tn_bf16", dY, dY.stride(0), X, X.stride(0), g, g.stride(0), M_out, N_in, T, 1, self.eng.s)

    def _colsum(self, a, name, T, N):
        lib.call("vr2_colsum_bf16", a, a.stride(0), self.state.grad[name], T, N, self.eng.s)

    def _keywork(self, pl):
        if getattr(pl, "d_keywork_window", None) is None:
            pl.d_keywork_window = _i32(vp.segment_key_work(pl.cu_window), self.eng.device)
            pl.d_keywork_full = _i32(vp.segment_key_work(pl.cu_full), self.eng.device)
        return pl.d_keywork_window, pl.d_keywork_full

    # ------------------------------------------------------------------ forward with saved activations
    def forward(self, pl, a0):
        """a0: patch rows bf16 [P][Kpad] in window order (Engine.encode_videos(rows_only=True)) -> video tokens f32 [P/4][out]
        in the plan's window order: the values Engine.encode_videos yields, with the activations the backward needs kept."""
        eng, W, v = self.eng, self.W, self.eng.cfg.vision
        b = eng.buf
        P = pl.n_patches
        Hv, hd, nh, D = v.hidden_size, v.head_dim, v.num_heads, v.depth
        Hp, Ip = pad64(Hv), pad64(v.intermediate_size)
        I16 = (v.intermediate_size + 15) // 16 * 16
        scale = hd ** -0.5
        sv = None
        if self.train_tower:
            x = b("tv.x", (P, Hv), torch.float32)
            self._gemm(a0, "patch_w", None, x, P, EPI_F32)
            cs = b("tv.cos", (P, hd // 2), torch.float32)
            sn = b("tv.sin", (P, hd // 2), torch.float32)
            lib.call("vr2_vision_rope_table", pl.d_pos, eng.vis_inv_freq, cs, sn, P, hd, eng.s)
            sv = dict(x_in=b("tv.x_in", (D, P, Hv), torch.float32), xn1=b("tv.xn1", (D, P, Hp), torch.bfloat16, zero=True),
                      qkv=b("tv.qkv", (D, P, 3 * Hv), torch.bfloat16), att=b("tv.att", (D, P, Hp), torch.bfloat16, zero=True),
                      lse=b("tv.lse", (D, P, nh), torch.float32), x_mid=b("tv.x_mid", (D, P, Hv), torch.float32),
                      xn2=b("tv.xn2", (D, P, Hp), torch.bfloat16, zero=True), gu=b("tv.gu", (D, P, 2 * I16), torch.bfloat16),
                      act=b("tv.act", (D, P, Ip), torch.bfloat16, zero=True), cos=cs, sin=sn, a0=a0)
            qkv32 = b("tv.qkv32", (P, 3 * Hv), torch.float32)
            for i in range(D):
                work = pl.d_work_full if i in v.fullatt_block_indexes else pl.d_work_window
                sv["x_in"][i].copy_(x)
                xn1, qkv, att = sv["xn1"][i], sv["qkv"][i], sv["att"][i]
                eng._norm(x, W[f"b{i}.norm1"], xn1, P, Hv, 1e-6)
                self._gemm(xn1, f"b{i}.qkv_w", W[f"b{i}.qkv_b"], qkv32, P, EPI_F32)
                kq, vq = qkv[:, Hv:], qkv[:, 2 * Hv:]
                lib.call("vr2_rope_qkv", qkv32, qkv32.stride(0), cs, sn, P, nh, nh, hd, qkv, qkv.stride(0), kq, vq, qkv.stride(0), 0, eng.s)
                lib.call("vr2_attn_tok", work, work.shape[0], qkv, qkv.stride(0), kq, vq, kq, vq, qkv.stride(0), att, att.stride(0),
                         nh, 1, hd, scale, sv["lse"][i], eng.s)
                self._gemm(att, f"b{i}.proj_w", W[f"b{i}.proj_b"], x, P, EPI_ADD_F32)
                sv["x_mid"][i].copy_(x)
                xn2, gu, act = sv["xn2"][i], sv["gu"][i], sv["act"][i]
                eng._norm(x, W[f"b{i}.norm2"], xn2, P, Hv, 1e-6)
                ngu, kgu = W.shape[f"b{i}.gu_w"]
                lib.call("vr2_gemm_swiglu_save", xn2, xn2.stride(0), W[f"b{i}.gu_w"], W[f"b{i}.gu_b"], act, act.stride(0), gu, gu.stride(0),
                         P, ngu, kgu, eng.s)
                self._gemm(act, f"b{i}.down_w", W[f"b{i}.down_b"], x, P, EPI_ADD_F32)
        else:
            x = eng._vision_tower(W, pl, a0)                           # frozen tower: nothing of it is kept
        # merger: ln_q -> fc1 (+ erf-GELU) -> fc2 over groups of merge_size^2 patches
        unit = v.spatial_merge_size ** 2
        Tv, mh = P // unit, Hv * unit
        xf = b("tv.xf", (P, Hv), torch.float32)
        xf.copy_(x)
        mn = b("tv.mn", (P, Hv), torch.bfloat16)
        eng._norm(xf, W["ln_q"], mn, P, Hv, 1e-6)
        mnv = mn.view(Tv, mh)
        z1 = b("tv.z1", (Tv, mh), torch.float32)
        self._gemm(mnv, "fc1_w", W["fc1_b"], z1, Tv, EPI_F32)
        h1 = b("tv.h1", (Tv, mh), torch.bfloat16)
        lib.call("vr2_gelu_fwd", z1, z1.stride(0), h1, h1.stride(0), Tv, mh, eng.s)
        out = b("tv.out", (Tv, v.out_hidden_size), torch.float32)
        self._gemm(h1, "fc2_w", W["fc2_b"], out, Tv, EPI_F32)
        self._saved = dict(pl=pl, sv=sv, xf=xf, mn=mn, z1=z1, h1=h1, P=P, Tv=Tv)
        return out

    # ------------------------------------------------------------------ backward (accumulates into VisionState.grad)
    def backward(self, d_out):
        """d_out f32 [Tv][out_hidden]: gradient of the loss w.r.t. the video tokens forward() returned (window order)."""
        s = self._saved
        assert s is not None, "VisionTrainer.backward() needs a preceding forward()"
        eng, W, st, v = self.eng, self.W, self.state, self.eng.cfg.vision
        b = eng.buf
        pl, sv, P, Tv = s["pl"], s["sv"], s["P"], s["Tv"]
        Hv, hd, nh, D = v.hidden_size, v.head_dim, v.num_heads, v.depth
        Hp, Ip = pad64(Hv), pad64(v.intermediate_size)
        I16 = (v.intermediate_size + 15) // 16 * 16
        unit = v.spatial_merge_size ** 2
        mh, Ho = Hv * unit, v.out_hidden_size
        rinv = b("tv.rinv", (P,), torch.float32)
        dx = b("tv.dx", (P, Hv), torch.float32)
        # ---- merger.  fc2 / fc1 gradients only when the merger trains; the input gradients always (they feed ln_q or the tower)
        dz2 = b("tv.dz2", (Tv, pad64(Ho)), torch.bfloat16, zero=True)
        lib.call("vr2_cast_rows_f32_bf16", d_out, d_out.stride(0), None, dz2, dz2.stride(0), Tv, Ho, Ho, eng.s)
        if self.train_merger:
            self._colsum(dz2, "fc2_b", Tv, Ho)
            self._wgrad(dz2, s["h1"], "fc2_w", Ho, mh, Tv)
        dh1 = b("tv.dh1", (Tv, mh), torch.bfloat16)
        self._gemm(dz2, "fc2_w_t", None, dh1, Tv, EPI_BF16)
        dz1 = b("tv.dz1", (Tv, mh), torch.bfloat16)
        lib.call("vr2_gelu_bwd", s["z1"], s["z1"].stride(0), dh1, dh1.stride(0), dz1, dz1.stride(0), Tv, mh, eng.s)
        if self.train_merger:
            self._colsum(dz1, "fc1_b", Tv, mh)
            self._wgrad(dz1, s["mn"].view(Tv, mh), "fc1_w", mh, mh, Tv)
        dmn = b("tv.dmn", (Tv, mh), torch.float32)
        self._gemm(dz1, "fc1_w_t", None, dmn, Tv, EPI_F32)
        dmn_rows = dmn.view(P, Hv)
        dxb = b("tv.dxb", (P, Hp), torch.bfloat16, zero=True)
        lib.call("vr2_rmsnorm_bwd", s["xf"], Hv, W["ln_q"], dmn_rows, Hv, dx, Hv, rinv, st.grad["ln_q"] if self.train_merger else None, P, Hv,
                 1e-6, 0, dxb if self.train_tower else None, dxb.stride(0), eng.s)
        if not self.train_tower:
            self._saved = None
            return
        # ---- ViT blocks, last to first
        kw_win, kw_full = self._keywork(pl)
        dact = b("tv.dact", (P, Ip), torch.bfloat16, zero=True)
        dgu = b("tv.dgu", (P, pad64(2 * I16)), torch.bfloat16, zero=True)      # K operand of the gu dgrad: padded to its tile
        dxn = b("tv.dxn", (P, Hp), torch.float32)                               # the transposed copies have pad64(Hv) output columns
        datt = b("tv.datt", (P, Hp), torch.bfloat16, zero=True)
        W128 = nh * 128
        q128, k128, v128 = (b("tv.%s128" % n, (P, W128), torch.bfloat16, zero=True) for n in "qkv")
        o128, do128 = b("tv.o128", (P, W128), torch.bfloat16, zero=True), b("tv.do128", (P, W128), torch.bfloat16, zero=True)
        dq128 = b("tv.dq128", (P, W128), torch.bfloat16)
        dk128, dv128 = b("tv.dk128", (P, W128), torch.float32), b("tv.dv128", (P, W128), torch.float32)
        rowc = b("tv.rowc", (2, nh, P), torch.float32)
        dq = b("tv.dq", (P, Hv), torch.bfloat16)
        dk, dvv = b("tv.dk", (P, Hv), torch.float32), b("tv.dv", (P, Hv), torch.float32)
        dqkv = b("tv.dqkv", (P, pad64(3 * Hv)), torch.bfloat16, zero=True)
        scale = hd ** -0.5

        def pad_heads(src, dst):     # [P][nh x hd] -> [P][nh x 128], the columns behind hd stay zero
            dst.view(P, nh, 128)[:, :, :hd].copy_(src.reshape(P, nh, hd))

        for i in reversed(range(D)):
            full = i in v.fullatt_block_indexes
            work = pl.d_work_full if full else pl.d_work_window
            kwork = kw_full if full else kw_win
            # MLP
            self._colsum(dxb, f"b{i}.down_b", P, Hv)
            self._gemm(dxb, f"b{i}.down_w_t", None, dact, P, EPI_BF16)
            self._wgrad(dxb, sv["act"][i], f"b{i}.down_w", Hv, Ip, P)
            gu = sv["gu"][i]
            lib.call("vr2_swiglu_bwd", gu, gu.stride(0), dact, dact.stride(0), dgu, dgu.stride(0), P, I16, eng.s)
            self._colsum(dgu, f"b{i}.gu_b", P, 2 * I16)
            self._gemm(dgu, f"b{i}.gu_w_t", None, dxn, P, EPI_F32)
            self._wgrad(dgu, sv["xn2"][i], f"b{i}.gu_w", 2 * I16, Hp, P)
            lib.call("vr2_rmsnorm_bwd", sv["x_mid"][i], Hv, W[f"b{i}.norm2"], dxn, dxn.stride(0), dx, Hv, rinv, st.grad[f"b{i}.norm2"], P, Hv, 1e-6, 1,
                     dxb, dxb.stride(0), eng.s)
            # attention
            self._colsum(dxb, f"b{i}.proj_b", P, Hv)
            self._gemm(dxb, f"b{i}.proj_w_t", None, datt, P, EPI_BF16)
            self._wgrad(dxb, sv["att"][i], f"b{i}.proj_w", Hv, Hp, P)
            qkv = sv["qkv"][i]
            pad_heads(qkv[:, :Hv], q128); pad_heads(qkv[:, Hv:2 * Hv], k128); pad_heads(qkv[:, 2 * Hv:], v128)
            pad_heads(sv["att"][i][:, :Hv], o128); pad_heads(datt[:, :Hv], do128)
            lib.call("vr2_attn_delta", o128, do128, W128, P, nh, 128, sv["lse"][i], rowc, eng.s)
            lib.call("vr2_attn_bwd_dq", work, work.shape[0], q128, W128, k128, v128, k128, v128, W128, do128, rowc, P, dq128, nh, 1, 128,
                     scale, eng.s)
            dk128.zero_(); dv128.zero_()
            lib.call("vr2_attn_bwd_dkv", kwork, kwork.shape[0], q128, W128, k128, v128, W128, do128, rowc, P, dk128, dv128, nh, 1, 128,
                     scale, eng.s)
            dq.view(P, nh, hd).copy_(dq128.view(P, nh, 128)[:, :, :hd])
            dk.view(P, nh, hd).copy_(dk128.view(P, nh, 128)[:, :, :hd])
            dvv.view(P, nh, hd).copy_(dv128.view(P, nh, 128)[:, :, :hd])
            lib.call("vr2_rope_bwd", dq, dq.stride(0), dk, dvv, dk.stride(0), sv["cos"], sv["sin"], P, nh, nh, hd, dqkv, dqkv.stride(0), eng.s)
            self._colsum(dqkv, f"b{i}.qkv_b", P, 3 * Hv)
            self._gemm(dqkv, f"b{i}.qkv_w_t", None, dxn, P, EPI_F32)
            self._wgrad(dqkv, sv["xn1"][i], f"b{i}.qkv_w", 3 * Hv, Hp, P)
            lib.call("vr2_rmsnorm_bwd", sv["x_in"][i], Hv, W[f"b{i}.norm1"], dxn, dxn.stride(0), dx, Hv, rinv, st.grad[f"b{i}.norm1"], P, Hv, 1e-6, 1,
                     dxb, dxb.stride(0), eng.s)
        # ---- patch embedding: dW += dx^T a0 (the pixels carry no gradient)
        a0 = sv["a0"]
        self._wgrad(dxb, a0, "patch_w", Hv, a0.shape[1], P)
        self._saved = None
