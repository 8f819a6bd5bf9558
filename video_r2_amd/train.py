"""Update half of the GRPO step on the MI355X engine: training forward (activations saved), hand-written backward, fused AdamW.

Replaces ``accelerator.backward(loss)`` + DeepSpeed ZeRO-3 ``optimizer.step()`` for the trainable LM (vision tower and merger are
frozen, /root/reference/train/scripts/train_grpo.sh:47-49; ``configure_llm`` train_grpo.py:59-64) -- SURVEY.md a17, K17, K18.

Formulation.  One prompt group = ONE unified row space of T = S + G*C rows: the S prompt rows (causal among themselves) followed
by the C completion rows of each of the G sequences (each sees the whole prompt + its own causal prefix).  This is exactly the
reference's batch of G copies of [prompt | completion] with the G identical prompt copies computed once; the backward sums the
prompt's K/V gradients over the G sequences, so the weight gradients equal the reference's (autograd through the full sequences).
With 288 GB of HBM nothing is recomputed: every layer keeps its inputs (~1.3 GB per layer at T = 9280), which replaces the
reference's gradient checkpointing (train_grpo.sh:74) -- that flag is accepted and ignored.

Mixed precision as the reference's bf16 autocast + fp32 master weights: GEMM operands bf16, accumulation / residual gradients
fp32, fp32 master parameters and Adam moments; the AdamW kernel re-emits the bf16 compute copies (tile-packed W for forward and
decode, tile-packed W^T for the dgrad GEMMs, row-major embedding table) in the same pass.
"""
from __future__ import annotations

import numpy as np
import torch

from . import dist as vdist
from . import lib
from .engine import EPI_BF16, EPI_F32, EPI_ADD_F32, Engine, _i32, _on_stream
from .host import rope_index as ri
from .host import vision_plan as vp
from .weights import Params, pack_tiles, pad64, unpack_tiles


class TrainState:
    """fp32 masters, gradients and Adam moments of the trainable LM tensors + the transposed packed copies for dgrad.

    The gradients live in ONE contiguous fp32 arena, laid out in the order the backward pass finishes them (lm_head + final
    norm, then layer L-1 ... layer 0, then the embedding rows): a bucket = one contiguous slice = everything one backward
    stage produced, so the data-parallel exchange is one all-reduce per bucket, issued the moment the stage's last weight
    gradient is enqueued and running on RCCL's stream under the rest of the backward (the single
    gradient exchange that replaces ZeRO-3's gathers / reduce-scatters)."""

    def __init__(self, text: Params, device, trainable=True):
        """trainable=False (--freeze_llm True under a trainable vision side): the LM keeps no masters, gradients or moments -- only the
        transposed copies its input gradients need on their way down to the vision tower."""
        self.text = text
        self.device = device
        self.trainable = trainable
        self.master, self.grad, self.m, self.v = {}, {}, {}, {}
        self.kind = {}            # 'mat' | 'rows' | 'vec'
        self.decay = {}
        self.t_name = {}
        cfg = text.cfg
        t = cfg.text
        L = t.num_hidden_layers
        mats = [f"l{i}.{n}" for i in range(L) for n in ("qkv_w", "o_w", "gu_w", "down_w")] + ["lm_head"]
        for name in mats:
            n, kp = text.shape[name]
            w = unpack_tiles(text[name], n, kp)                      # [N][Kp] bf16 (K padding columns are zero)
            if trainable:
                self._add(name, w.float(), "mat", True)
            wt = torch.zeros((kp, pad64(n)), dtype=torch.bfloat16, device=device)
            wt[:, :n] = w.t()
            text.t[name + "_t"] = pack_tiles(wt)
            text.shape[name + "_t"] = (kp, pad64(n))
        self.tied = t.tie_word_embeddings
        self.step_count = 0
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=device)
        self.coef = torch.ones(1, dtype=torch.float32, device=device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=device)
        self.arena, self.buckets = torch.zeros(0, dtype=torch.float32, device=device), []
        if not trainable:
            return
        self._add("embed", text["embed"].float(), "rows", True)
        vecs = [f"l{i}.{n}" for i in range(L) for n in ("ln1", "ln2", "qkv_b")] + ["norm"]
        for name in vecs:
            self._add(name, text[name], "vec", False)                # the fp32 vector IS its own master (updated in place)
        # ---- gradient arena in backward-completion order; buckets = (key, first element, one-past-last element)
        order = [("head", ["lm_head", "norm"])]
        order += [(l, [f"l{l}.{n}" for n in ("down_w", "gu_w", "ln2", "o_w", "qkv_b", "qkv_w", "ln1")]) for l in reversed(range(L))]
        order += [("embed", ["embed"])]
        total = sum(self.master[n].numel() for _, names in order for n in names)
        assert total == sum(m.numel() for m in self.master.values())
        self.arena = torch.zeros(total, dtype=torch.float32, device=device)
        self.buckets = []
        at = 0
        for key, names in order:
            first = at
            for n in names:
                k = self.master[n].numel()
                self.grad[n] = self.arena[at:at + k].view(self.master[n].shape)
                at += k
            self.buckets.append((key, first, at))

    def _add(self, name, master, kind, decay):
        self.master[name] = master.contiguous()
        self.m[name] = torch.zeros_like(self.master[name])
        self.v[name] = torch.zeros_like(self.master[name])
        self.kind[name] = kind
        self.decay[name] = decay     # HF Trainer: no weight decay on biases and norm weights

    def zero_grad(self):
        self.arena.zero_()

    def n_params(self):
        return sum(m.numel() for m in self.master.values())


class Trainer:
    """Training forward / backward / optimiser step on an Engine (shares its stream and workspaces)."""
    KEEP_LOGITS_BYTES = 8 << 30      # forward() keeps a prompt group's fp32 logits for backward() up to this size

    def __init__(self, engine: Engine, text: Params, grad_exchange="fp32", vision_trainer=None, train_lm=True, lora=None):
        """grad_exchange: wire format of the data-parallel gradient all-reduce.  "fp32" (default) exchanges the fp32 arena as it
        is (30.5 GB per step at 7B: the exact mean of the ranks' fp32 gradients).  "bf16" rounds every bucket to bfloat16 on the
        way out, sums in bfloat16 on the wire and widens the result back into the arena: 15.2 GB per step -- the figure of the
        reference's bf16 ZeRO-3 reduce-scatter -- at a bf16 rounding per summand; every rank receives the same
        reduced values either way, so replicas stay bit-identical.
        vision_trainer: a train_vision.VisionTrainer when the vision tower and / or the merger train (--freeze_vision_tower False /
        --freeze_merger False): forward() then runs the tower with saved activations from the patch rows it is handed, backward() hands
        the input gradient at the video-token rows down to it, and optimizer_step() clips and updates both parameter sets together.
        train_lm=False (--freeze_llm True; needs a vision_trainer or LoRA adapters): the LM's backward computes input gradients only -- no weight
        gradients, no masters or moments for the LM -- and the optimiser step covers the vision tensors alone."""
        if not train_lm and vision_trainer is None and lora is None:
            raise ValueError("Trainer(train_lm=False) without a vision_trainer or LoRA adapters: nothing to train")
        if lora is not None and train_lm:
            raise ValueError("LoRA adapters go on a frozen LM (train_lm=False), as the reference requires (train_sft.py:88-89)")
        if grad_exchange not in ("fp32", "bf16"):
            raise ValueError(f"grad_exchange must be fp32 or bf16, got {grad_exchange!r}")
        self.eng = engine
        self.text = text
        self.train_lm = bool(train_lm)
        self.state = TrainState(text, engine.device, trainable=self.train_lm)
        self.grad_exchange = grad_exchange
        self.vis = vision_trainer
        self.lora = lora          # lora.LoraTrainer: adapters on the LM's linear layers (the LM itself frozen)
        self._wire = {}           # bf16 mirrors of the gradient arenas (allocated at the first bf16 exchange)
        self._saved = None
        self._works = {}          # bucket key -> in-flight all-reduce of this optimiser step
        self._watch = None        # deadline on the in-flight buckets (dist.ExchangeWatch)

    @property
    def device(self):
        return self.eng.device

    @property
    def stream(self):
        return self.eng.stream

    # ------------------------------------------------------------------ helpers
    def _gemm(self, A, wname, bias, C, M, epi):
        self.eng._gemm(A, self.text, wname, bias, C, M, epi)

    def _wgrad(self, dY, X, name, M_out, N_in, T):
        if not self.train_lm:
            return
        g = self.state.grad[name]
        lib.call("vr2_gemm_tn_bf16", dY, dY.stride(0), X, X.stride(0), g, g.stride(0), M_out, N_in, T, 1, self.eng.s)

    def _cast(self, src, dst, T, H):
        lib.call("vr2_cast_rows_f32_bf16", src, src.stride(0), None, dst, dst.stride(0), T, H, H, self.eng.s)

    # ------------------------------------------------------------------ data-parallel gradient exchange
    def _exchange(self, key):
        """SUM all-reduce of one arena bucket, asynchronous: torch's process group (RCCL on the "nccl" backend) orders it
        after everything already enqueued on the engine stream and runs it on its own stream, under the kernels enqueued
        next.  The 1/world of the mean is folded into the clip coefficient (optimizer_step)."""
        if vdist.world_size() == 1 or key in self._works:
            return
        import torch.distributed as dist
        arena, first, last = self._bucket(key)
        buf = arena[first:last]
        if self.grad_exchange == "bf16":
            wire = self._wire.get(id(arena))
            if wire is None:
                wire = self._wire[id(arena)] = torch.empty(arena.numel(), dtype=torch.bfloat16, device=arena.device)
            buf = wire[first:last]
            buf.copy_(arena[first:last])                            # one rounding per summand, on the engine stream
        self._works[key] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        if self._watch is None:
            self._watch = vdist.ExchangeWatch()
        self._watch.add({"head": "head (lm_head + final norm)", "embed": "embedding rows", "vision": "vision tower / merger", "lora": "LoRA adapters"}.get(key, f"layer {key}"),
                        self._works[key], buf.numel() * buf.element_size())

    def _bucket(self, key):
        """(arena, first element, one-past-last element) of a gradient bucket: one of the LM's backward stages, or the vision side's"""
        for st in ([self.state] + ([self.vis.state] if self.vis is not None else []) + ([self.lora] if self.lora is not None else [])):
            for k, first, last in st.buckets:
                if k == key:
                    return st.arena, first, last
        raise KeyError(key)

    def _bucket_keys(self):
        return [k for k, _, _ in self.state.buckets] + (["vision"] if self.vis is not None else []) + (["lora"] if self.lora is not None else [])

    # ------------------------------------------------------------------ forward with saved activations
    @_on_stream
    def forward(self, prompt_ids, video_embeds, plan, second_per_grid_ts, completion_ids, temperature, lens=None, vision_rows=None):
        """lens (host ints, one per sequence): train on the first lens[g] tokens of completion g only -- the rows behind a
        completion's first EOS carry no loss (grpo_trainer.py:793-806) and, being causal successors, influence no other row, so
        they are left out of the row space: T = S + sum(lens) rows instead of S + G*C, log-probs 0 at the skipped positions.
        vision_rows (with a vision_trainer): the prompt's patch rows bf16 [P][Kpad] from Engine.encode_videos(rows_only=True); the
        vision tokens are then computed HERE, with saved activations, instead of taken from video_embeds."""
        eng, W, cfg = self.eng, self.text, self.eng.cfg
        if self.vis is not None and vision_rows is not None and plan is not None:
            video_embeds = self.vis.forward(plan, vision_rows)
        t = cfg.text
        H, hd, nq, nkv, I, V, L = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size, t.vocab_size, t.num_hidden_layers
        I16 = (I + 15) // 16 * 16
        nqkv = (nq + 2 * nkv) * hd
        # row space, positions, attention work list and head gather: Engine.unified_layout, shared with Engine.score_unified (the
        # no-grad pass of the reference / old-policy log-probs) so that the two passes run the same launches on the same rows
        u = eng.unified_layout(prompt_ids, plan, second_per_grid_ts, video_embeds, completion_ids, lens)
        if u["R"] == 0:
            raise ValueError("forward(): every completion has length 0 -- nothing to train on")
        S, G, C, R, T, lay = u["S"], u["G"], u["C"], u["R"], u["T"], u["lay"]
        tok, all_ids, d_work, d_sel = u["tok"], u["all_ids"], u["work"], u["sel"]
        b = eng.buf
        x, cos, sin = eng.unified_inputs(W, u, video_embeds, plan, "t")
        kwork = _i32(vp.key_work(S, G, C, group=t.group, lens=None if lay is None else u["seq_len"]), eng.device)
        lo = self.lora
        fid = lo.begin_forward() if lo is not None else 0
        xk = lo.extra_k("gu") if lo is not None else 0      # LoRA: the gate/up adapters' t columns ride behind the normed row
        sv = dict(x_in=b("t.x_in", (L, T, H), torch.float32), xn1=b("t.xn1", (L, T, H), torch.bfloat16),
                  q=b("t.q", (L, T, nq * hd), torch.bfloat16), k=b("t.k", (L, T, nkv * hd), torch.bfloat16),
                  v=b("t.v", (L, T, nkv * hd), torch.bfloat16), att=b("t.att", (L, T, nq * hd), torch.bfloat16),
                  lse=b("t.lse", (L, T, nq), torch.float32), x_mid=b("t.x_mid", (L, T, H), torch.float32),
                  xn2=b("t.xn2", (L, T, H + xk), torch.bfloat16), gu=b("t.gu", (L, T, 2 * I16), torch.bfloat16),
                  act=b("t.acts", (L, T, I16), torch.bfloat16))   # SwiGLU output kept for the down-projection wgrad (9.8 GB at 7B)
        if lo is not None:
            sv.update(t_qkv=b("t.lora_qkv", (L, T, lo.extra_k("qkv")), torch.bfloat16), t_o=b("t.lora_o", (L, T, lo.extra_k("o")), torch.bfloat16),
                      t_down=b("t.lora_down", (L, T, lo.extra_k("down")), torch.bfloat16))
        qkv32 = b("t.qkv32", (T, nqkv), torch.float32)
        scale = hd ** -0.5
        for l in range(L):
            act = sv["act"][l]
            sv["x_in"][l].copy_(x)
            xn1 = sv["xn1"][l]
            eng._norm(x, W[f"l{l}.ln1"], xn1, T, H, t.rms_norm_eps)
            self._gemm(xn1, f"l{l}.qkv_w", W[f"l{l}.qkv_b"], qkv32, T, EPI_F32)
            if lo is not None:
                lo.project(fid, l, "qkv", xn1, sv["t_qkv"][l], T)
                lo.join(l, "qkv", sv["t_qkv"][l], qkv32, T)
            q, k, v = sv["q"][l], sv["k"][l], sv["v"][l]
            lib.call("vr2_rope_qkv", qkv32, qkv32.stride(0), cos, sin, T, nq, nkv, hd, q, q.stride(0), k, v, k.stride(0), 0, eng.s)
            att = sv["att"][l]
            lib.call("vr2_attn_tok", d_work, d_work.shape[0], q, q.stride(0), k, v, k, v, k.stride(0), att, att.stride(0), nq,
                     t.group, hd, scale, sv["lse"][l], eng.s)
            self._gemm(att, f"l{l}.o_w", None, x, T, EPI_ADD_F32)
            if lo is not None:
                lo.project(fid, l, "o", att, sv["t_o"][l], T)
                lo.join(l, "o", sv["t_o"][l], x, T)
            sv["x_mid"][l].copy_(x)
            xn2 = sv["xn2"][l]
            eng._norm(x, W[f"l{l}.ln2"], xn2, T, H, t.rms_norm_eps)
            gu = sv["gu"][l]
            ngu, kgu = W.shape[f"l{l}.gu_w"]
            gu_w = W[f"l{l}.gu_w"]
            if lo is not None:                 # [xn2 | t] x [W | s B]^T: the adapters join before the SwiGLU epilogue
                lo.project(fid, l, "gu", xn2[:, :H], xn2[:, H:], T)
                gu_w, (ngu, kgu) = lo.W[f"l{l}.gu_we"], lo.W.shape[f"l{l}.gu_we"]
            lib.call("vr2_gemm_swiglu_save", xn2, xn2.stride(0), gu_w, None, act, act.stride(0), gu, gu.stride(0), T,
                     ngu, kgu, eng.s)
            self._gemm(act, f"l{l}.down_w", None, x, T, EPI_ADD_F32)
            if lo is not None:
                lo.project(fid, l, "down", act, sv["t_down"][l], T)
                lo.join(l, "down", sv["t_down"][l], x, T)
        # log-prob head: row (g, j) is predicted by the prompt's last row (j == 0) or completion row (g, j-1)
        hs = b("t.hs", (R, H), torch.float32)
        torch.index_select(x, 0, u["pred"], out=hs)
        hn = b("t.hn", (R, H), torch.bfloat16)
        logp = b("t.logp", (R,), torch.float32)
        lse_out = b("t.lse_out", (R,), torch.float32)
        # the group's logits stay in memory for the backward when they fit KEEP_LOGITS_BYTES (7B, G x C = 8192 rows: 5 GB): the
        # head backward then reads them instead of multiplying hn by the lm_head a second time
        keep = R * V * 4 <= self.KEEP_LOGITS_BYTES
        lg_all = b("t.logits", (R, V), torch.float32) if keep else None
        eng.head_logps(W, hs, u, temperature, logp, hn=hn, logits_all=lg_all, lse_out=lse_out, allow_f8=False)
        self._saved = dict(fid=fid, logits=lg_all, sv=sv, S=S, G=G, C=C, T=T, R=R, cos=cos, sin=sin, work=d_work, kwork=kwork, ids=all_ids, hs=hs, hn=hn, tok=tok,
                           lse_out=lse_out, temperature=temperature, has_video=video_embeds is not None, sel=d_sel,
                           vis_dst=(_i32(u["vid_pos"][plan.window_index], eng.device).long()
                                    if (self.vis is not None and vision_rows is not None and plan is not None) else None),
                           pred=None if lay is None else u["pred"], first=None if lay is None else u["first"])
        return eng.unified_logps_out(u, logp)

    # ------------------------------------------------------------------ backward (accumulates into TrainState.grad)
    @_on_stream
    def backward(self, dlogp, exchange=False):
        """exchange=True on the LAST backward of an optimiser step (gradients accumulate over the step's prompt groups): every
        stage's bucket is all-reduced as soon as its weight gradients are enqueued, overlapping the rest of this backward."""
        eng, W, cfg, st = self.eng, self.text, self.eng.cfg, self.state
        exchange = exchange and vdist.world_size() > 1 and not st.tied     # tied lm_head / embedding: summed first, at the step
        lm_exchange = exchange and self.train_lm
        gr = (lambda n: st.grad[n]) if self.train_lm else (lambda n: None)   # frozen LM: the norm kernels skip their weight gradient
        s = self._saved
        assert s is not None, "backward() needs a preceding forward()"
        t = cfg.text
        H, hd, nq, nkv, I, V, L = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size, t.vocab_size, t.num_hidden_layers
        I16 = (I + 15) // 16 * 16
        nqkv = (nq + 2 * nkv) * hd
        S, G, C, T, R, sv = s["S"], s["G"], s["C"], s["T"], s["R"], s["sv"]
        lo, fid = self.lora, s["fid"]
        b = eng.buf
        dl = dlogp.reshape(-1).to(torch.float32).contiguous()
        if s["sel"] is not None:
            dl = dl.index_select(0, s["sel"]).contiguous()          # packed rows: the gradient of the positions that were computed
        # ---- log-prob head
        Vp = W.shape["lm_head_t"][1]
        rc = min(4096, R)        # rows per logits chunk: 4096 x 152064 fp32 = 2.5 GB; keeps the head GEMMs on 256-row tiles
        lg_all = s.get("logits")
        lg = None if lg_all is not None else b("lm.chunk_logits", (rc, V), torch.float32)
        dz = b("t.dz", (rc, Vp), torch.bfloat16)
        dhn = b("t.dhn", (R, H), torch.float32)
        for r0 in range(0, R, rc):
            n = min(rc, R - r0)
            if lg_all is not None:
                lg = lg_all[r0:r0 + n]                                   # kept by forward()
            else:
                self._gemm(s["hn"][r0:r0 + n], "lm_head", None, lg, n, EPI_F32)
            lib.call("vr2_dlogits", lg, lg.stride(0), n, V, s["tok"][r0:r0 + n], s["lse_out"][r0:r0 + n], dl[r0:r0 + n],
                     s["temperature"], dz, dz.stride(0), eng.s)
            self._gemm(dz, "lm_head_t", None, dhn[r0:r0 + n], n, EPI_F32)
            self._wgrad(dz, s["hn"][r0:r0 + n], "lm_head", V, H, n)
        dhs = b("t.dhs", (R, H), torch.float32)
        rinv = b("t.rinv", (max(T, R),), torch.float32)
        lib.call("vr2_rmsnorm_bwd", s["hs"], H, W["norm"], dhn, H, dhs, H, rinv, gr("norm"), R, H, t.rms_norm_eps, 0, None, 0, eng.s)
        if lm_exchange:
            self._exchange("head")
        dx = b("t.dx", (T, H), torch.float32)
        dx.zero_()
        if s["sel"] is None:
            dv_, dc = dhs.view(G, C, H), dx[S:].view(G, C, H)
            dc[:, :-1].copy_(dv_[:, 1:])
            dx[S - 1].copy_(dv_[:, 0].sum(0))
        else:   # head row r came from x row pred[r]: unique rows except the prompt's last one, which fed every sequence's first row
            dx.index_copy_(0, s["pred"], dhs)
            dx[S - 1].copy_(dhs.index_select(0, s["first"]).sum(0))
        dxb = b("t.dxb", (T, H), torch.bfloat16)
        dact = b("t.dact", (T, I16), torch.bfloat16)
        dgu = b("t.dgu", (T, 2 * I16), torch.bfloat16)
        dxn = b("t.dxn", (T, H), torch.float32)
        datt = b("t.datt", (T, nq * hd), torch.bfloat16)
        rowc = b("t.rowc", (2, nq, T), torch.float32)
        dq = b("t.dq", (T, nq * hd), torch.bfloat16)
        dk = b("t.dk32", (T, nkv * hd), torch.float32)
        dvv = b("t.dv32", (T, nkv * hd), torch.float32)
        dqkv = b("t.dqkv", (T, nqkv), torch.bfloat16)
        scale = hd ** -0.5
        self._cast(dx, dxb, T, H)          # after this, every update of dx (rmsnorm_bwd) also refreshes its bf16 copy dxb
        for l in reversed(range(L)):
            # ---- MLP
            self._gemm(dxb, f"l{l}.down_w_t", None, dact, T, EPI_BF16)
            if lo is not None:
                lo.backward(fid, l, "down", dxb, sv["act"][l], sv["t_down"][l], dact, T)
            gu = sv["gu"][l]
            self._wgrad(dxb, sv["act"][l], f"l{l}.down_w", H, I16, T)
            lib.call("vr2_swiglu_bwd", gu, gu.stride(0), dact, dact.stride(0), dgu, dgu.stride(0), T, I16, eng.s)
            self._gemm(dgu, f"l{l}.gu_w_t", None, dxn, T, EPI_F32)
            if lo is not None:
                lo.backward(fid, l, "gu", dgu, sv["xn2"][l][:, :H], sv["xn2"][l][:, H:], dxn, T)
            self._wgrad(dgu, sv["xn2"][l], f"l{l}.gu_w", 2 * I16, H, T)
            lib.call("vr2_rmsnorm_bwd", sv["x_mid"][l], H, W[f"l{l}.ln2"], dxn, H, dx, H, rinv, gr(f"l{l}.ln2"), T, H,
                     t.rms_norm_eps, 1, dxb, dxb.stride(0), eng.s)
            # ---- attention
            self._gemm(dxb, f"l{l}.o_w_t", None, datt, T, EPI_BF16)
            att, q, k, v, lse = sv["att"][l], sv["q"][l], sv["k"][l], sv["v"][l], sv["lse"][l]
            if lo is not None:
                lo.backward(fid, l, "o", dxb, att, sv["t_o"][l], datt, T)
            self._wgrad(dxb, att, f"l{l}.o_w", H, nq * hd, T)
            lib.call("vr2_attn_delta", att, datt, att.stride(0), T, nq, hd, lse, rowc, eng.s)
            lib.call("vr2_attn_bwd_dq", s["work"], s["work"].shape[0], q, q.stride(0), k, v, k, v, k.stride(0), datt, rowc, T, dq,
                     nq, t.group, hd, scale, eng.s)
            dk.zero_(); dvv.zero_()
            lib.call("vr2_attn_bwd_dkv", s["kwork"], s["kwork"].shape[0], q, q.stride(0), k, v, k.stride(0), datt, rowc, T, dk, dvv,
                     nq, t.group, hd, scale, eng.s)
            lib.call("vr2_rope_bwd", dq, dq.stride(0), dk, dvv, dk.stride(0), s["cos"], s["sin"], T, nq, nkv, hd, dqkv, dqkv.stride(0), eng.s)
            if self.train_lm:
                lib.call("vr2_colsum_bf16", dqkv, dqkv.stride(0), st.grad[f"l{l}.qkv_b"], T, nqkv, eng.s)
            self._gemm(dqkv, f"l{l}.qkv_w_t", None, dxn, T, EPI_F32)
            if lo is not None:
                lo.backward(fid, l, "qkv", dqkv, sv["xn1"][l], sv["t_qkv"][l], dxn, T)
            self._wgrad(dqkv, sv["xn1"][l], f"l{l}.qkv_w", nqkv, H, T)
            lib.call("vr2_rmsnorm_bwd", sv["x_in"][l], H, W[f"l{l}.ln1"], dxn, H, dx, H, rinv, gr(f"l{l}.ln1"), T, H,
                     t.rms_norm_eps, 1, dxb, dxb.stride(0), eng.s)
            if lm_exchange:
                self._exchange(l)
        # ---- embedding rows (video placeholder rows carry frozen vision features: no gradient)
        if self.train_lm:
            skip = _i32([cfg.video_token_id if s["has_video"] else -1], eng.device)
            ge = st.grad["embed"]
            lib.call("vr2_embed_bwd", s["ids"], skip, dx, dx.stride(0), ge, ge.stride(0), T, H, eng.s)
        if lm_exchange:
            self._exchange("embed")
        if s.get("vis_dst") is not None:
            # the video-token rows of the LM input carry the vision side's output: their input gradient, in the window order the
            # merger produced them, goes down through the merger (and the tower)
            self.vis.backward(dx.index_select(0, s["vis_dst"]))
            if exchange:
                self._exchange("vision")
        if lo is not None and exchange:
            lo.finish_grads()
            self._exchange("lora")
        self._saved = None

    # ------------------------------------------------------------------ checkpoint resume
    @_on_stream
    def refresh_compute_copies(self):
        """Re-derive every bf16 compute copy (packed W, packed W^T, embedding rows) from the fp32 masters."""
        for gi, (st, W) in enumerate([(self.state, self.text)] + ([(self.vis.state, self.vis.W)] if self.vis is not None else [])):
            for name, p in st.master.items():
                if st.kind[name] == "mat":
                    src = st.master["embed"] if (gi == 0 and st.tied and name == "lm_head") else p
                    n, kp = W.shape[name]
                    w = torch.zeros((n, kp), dtype=torch.bfloat16, device=src.device)
                    w[:, :src.shape[1]] = src.to(torch.bfloat16)
                    W.t[name].copy_(pack_tiles(w))
                    kt, npad = W.shape[name + "_t"]
                    wt = torch.zeros((kt, npad), dtype=torch.bfloat16, device=src.device)
                    wt[:, :n] = w.t()
                    W.t[name + "_t"].copy_(pack_tiles(wt))
                elif st.kind[name] == "rows":
                    W.t["embed"].copy_(p.to(torch.bfloat16))

    # ------------------------------------------------------------------ optimiser step
    @_on_stream
    def optimizer_step(self, lr, weight_decay=0.01, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_norm=1.0, grad_scale=1.0,
                       vision_lr=None, merger_lr=None):
        """All-reduce (mean) the gradients over ranks, clip by global norm, AdamW on the fp32 masters, re-emit bf16 copies,
        zero the gradients.  grad_scale = 1 / (gradient accumulation count).  Returns the pre-clip global norm (0-d tensor).
        vision_lr / merger_lr: the SFT trainer's parameter groups (sft_trainer.py:50-110 of the reference: names with "visual" and
        without "merger" / names with "merger"); None = the common lr."""
        eng, st, W = self.eng, self.state, self.text
        if st.tied and self.train_lm:
            st.grad["embed"].add_(st.grad["lm_head"])        # one shared parameter (Qwen2.5-VL-3B)
        world = vdist.world_size()
        lo = self.lora
        if lo is not None:
            lo.finish_grads()
        if world > 1:
            for key in self._bucket_keys():                   # whatever the last backward did not already put on the wire
                self._exchange(key)
            for key, wk in self._works.items():
                wk.wait()                                     # the engine stream waits for RCCL's; the host does not
                if self.grad_exchange == "bf16":              # widen the reduced bucket back into the fp32 arena
                    arena, first, last = self._bucket(key)
                    arena[first:last].copy_(self._wire[id(arena)][first:last])
            self._works = {}
            grad_scale = grad_scale / world                   # SUM -> mean, folded into the clip coefficient
        groups = [(st, W)] + ([(self.vis.state, self.vis.W)] if self.vis is not None else [])
        names = [(g_, n_) for g_, (gs, _) in enumerate(groups) for n_ in gs.grad if not (g_ == 0 and st.tied and n_ == "lm_head")]
        slot_f = int(lib.load().vr2_sumsq_slot_floats())
        n_slots = len(names) + (len(lo.grad) if lo is not None else 0)
        if st.sumsq.numel() != n_slots * slot_f:
            st.sumsq = torch.zeros(n_slots * slot_f, dtype=torch.float32, device=st.sumsq.device)
        for slot, (g_, name) in enumerate(names):   # per-tensor slots of per-block partials, summed in a fixed order: no atomics
            g = groups[g_][0].grad[name]
            lib.call("vr2_sumsq_f32", g, g.numel(), st.sumsq, slot, eng.s)
        if lo is not None:
            for j, g in enumerate(lo.grad.values()):
                lib.call("vr2_sumsq_f32", g, g.numel(), st.sumsq, len(names) + j, eng.s)
        lib.call("vr2_clip_coef", st.sumsq, n_slots, float(max_grad_norm if max_grad_norm else 0.0), float(grad_scale), st.coef,
                 st.gnorm, eng.s)
        st.step_count += 1
        for g_, (gs, GW) in enumerate(groups):
            for name, p in gs.master.items():
                if g_ == 0 and st.tied and name == "lm_head":
                    continue
                g, m, v = gs.grad[name], gs.m[name], gs.v[name]
                wd = weight_decay if gs.decay[name] else 0.0
                kind = gs.kind[name]
                lr_all = lr
                if g_ > 0 and groups[g_][0] is getattr(self.vis, "state", None):
                    group_lr = merger_lr if name in ("fc1_w", "fc2_w", "ln_q", "fc1_b", "fc2_b") else vision_lr
                    lr = lr_all if group_lr is None else group_lr
                if kind == "mat":
                    n, kp = GW.shape[name]
                    kt, npad = GW.shape[name + "_t"]
                    lib.call("vr2_adamw", p, g, m, v, p.shape[0], p.shape[1], lr, beta1, beta2, eps, wd, st.step_count, st.coef,
                             GW[name], kp, GW[name + "_t"], npad, None, 0, eng.s)
                elif kind == "rows":
                    n, k = p.shape
                    if st.tied:
                        kp = GW.shape["lm_head"][1]
                        lib.call("vr2_adamw", p, g, m, v, n, k, lr, beta1, beta2, eps, wd, st.step_count, st.coef, GW["lm_head"], kp,
                                 GW["lm_head_t"], GW.shape["lm_head_t"][1], GW["embed"], GW["embed"].stride(0), eng.s)
                    else:
                        lib.call("vr2_adamw", p, g, m, v, n, k, lr, beta1, beta2, eps, wd, st.step_count, st.coef, None, 0, None, 0,
                                 GW["embed"], GW["embed"].stride(0), eng.s)
                else:
                    lib.call("vr2_adamw", p, g, m, v, 1, p.numel(), lr, beta1, beta2, eps, wd, st.step_count, st.coef, None, 0, None, 0,
                             None, 0, eng.s)
                lr = lr_all
        if lo is not None:      # the adapters: plain fp32 tensors (weight decay applies, HF's no-decay set is biases and norms)
            for k, p in lo.master.items():
                lib.call("vr2_adamw", p, lo.grad[k], lo.m[k], lo.v[k], 1, p.numel(), lr, beta1, beta2, eps, weight_decay, st.step_count, st.coef,
                         None, 0, None, 0, None, 0, eng.s)
            lo.refresh()
            lo.zero_grad()
        if self.vis is not None:
            self.vis.state.zero_grad()
        st.zero_grad()
        if self._watch is not None and world > 1:
            # the deadline thread keeps polling the step's works until they report complete; nothing to do here: clear() would
            # only be safe after a host sync, which the caller's read of the gradient norm provides -- completed works drop out
            pass
        return st.gnorm    # captured decode graphs stay valid: the weight buffers are updated in place


def cosine_lr(base_lr, step, total_steps, warmup_steps=0):
    """HF ``get_cosine_schedule_with_warmup`` (lr_scheduler_type cosine, train_grpo.sh:71)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + np.cos(np.pi * prog)))
