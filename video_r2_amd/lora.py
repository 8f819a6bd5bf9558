"""LoRA adapters on the LM's linear layers (`--lora_enable True`: the shipped SFT recipe, train_sft.sh:30-37 of the reference).

What the reference builds with peft (/root/reference/train/src/train/train_sft.py:153-169, train_grpo.py:137-149):
`LoraConfig(r, lora_alpha, target_modules = every nn.Linear outside lm_head / embed_tokens / visual, lora_dropout, bias="none")` ->
each adapted layer computes  base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r)  with lora_A ~ kaiming-uniform(a = sqrt 5)
(= U(-1/sqrt(in), 1/sqrt(in))) and lora_B = 0; only lora_A / lora_B train (the LM itself is frozen, train_sft.py:88-89).

Here the seven adapters of a decoder layer follow the engine's fused projections:

    site   input          adapters            fused A [r', K]        fused s*B [N, r'] (block structure)
    qkv    ln1(x)         q_proj k_proj v_proj  rows [A_q; A_k; A_v]   q rows x A_q's columns, k rows x A_k's, v rows x A_v's
    o      attention out  o_proj               A_o                    B_o
    gu     ln2(x)         gate_proj up_proj    rows [A_g; A_u]        interleaved gate rows x A_g's columns, up rows x A_u's
    down   SwiGLU out     down_proj            A_d                    B_d

(r' = adapters x pad64(r); padding rows / columns and the off-block entries are zero and stay zero.)  Forward: t = dropout(x) A^T
(one narrow GEMM), then the branch joins the base projection in fp32 -- as a second GEMM accumulating into the projection's fp32
output (qkv, o, down) or, for gate/up whose SwiGLU epilogue needs the sum BEFORE the activation, as extra K columns of the
projection itself: [xn | t] x [W | s B]^T.  With B = 0 both forms add exact zeros, so the adapted forward starts bit-identical to
the base model's.  Backward: dt = dY (s B), dA += dt^T dropout(x), dB += s dY^T t, and the input gradient gets
mask o (dt A) / (1 - p) on top of the frozen base layer's dY W (vr2_dropout_add redraws the forward's mask from its seed).

The fused fp32 masters are what AdamW updates (weight decay applies: HF's no-decay set is biases and norms only); `save_adapter`
writes them out module by module in peft's layout (adapter_model.safetensors + adapter_config.json), `merged_state_dict` folds
W + s B A into the base weights (merge_and_unload, train/src/merge_lora_weights.py of the reference).
"""
from __future__ import annotations

import json
import math
import os

import torch

from . import lib
from .engine import EPI_BF16, EPI_F32, EPI_ADD_F32, Engine
from .weights import Params, pack_tiles, pad64

SITES = ("qkv", "o", "gu", "down")
HF_ORDER = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


class LoraTrainer:
    def __init__(self, engine: Engine, text: Params, rank=64, alpha=16, dropout=0.05, num_modules=-1, seed=42):
        self.eng, self.text = engine, text
        self.r, self.alpha, self.p = int(rank), float(alpha), float(dropout)
        self.scaling = self.alpha / self.r
        self.rp = pad64(self.r)
        self.seed = int(seed)
        cfg = engine.cfg
        t = cfg.text
        self.L = t.num_hidden_layers
        H, hd, nq, nkv, I = t.hidden_size, t.head_dim, t.num_attention_heads, t.num_key_value_heads, t.intermediate_size
        a, c = nq * hd, (nq + nkv) * hd
        dev = engine.device
        names = [f"model.layers.{i}.{m}" for i in range(self.L) for m in HF_ORDER]
        self.enabled = set(names[-num_modules:] if num_modules > 0 else names)      # find_target_linear_names(num_lora_modules)
        gate_rows = torch.tensor([32 * (j // 16) + j % 16 for j in range(I)], device=dev)
        # site -> [(HF module, in features, out row indices in the fused projection)]
        self.layout = {
            "qkv": [("self_attn.q_proj", H, torch.arange(0, a, device=dev)), ("self_attn.k_proj", H, torch.arange(a, c, device=dev)),
                    ("self_attn.v_proj", H, torch.arange(c, (nq + 2 * nkv) * hd, device=dev))],
            "o": [("self_attn.o_proj", a, torch.arange(0, H, device=dev))],
            "gu": [("mlp.gate_proj", H, gate_rows), ("mlp.up_proj", H, gate_rows + 16)],
            "down": [("mlp.down_proj", I, torch.arange(0, H, device=dev))],
        }
        self.base = {"qkv": "qkv_w", "o": "o_w", "gu": "gu_w", "down": "down_w"}
        self.W = Params(cfg, dev)                      # packed bf16 compute copies of the adapters
        self.master, self.m, self.v, self.grad, self.mask = {}, {}, {}, {}, {}
        g = torch.Generator().manual_seed(self.seed)
        sizes = []
        for l in range(self.L):
            for site in SITES:
                n_out, kin = text.shape[f"l{l}.{self.base[site]}"]
                rr = len(self.layout[site]) * self.rp
                A = torch.zeros((rr, kin), dtype=torch.float32)
                mA = torch.zeros((rr, kin), dtype=torch.float32)
                mB = torch.zeros((n_out, rr), dtype=torch.float32, device=dev)
                for j, (mod, fin, rows) in enumerate(self.layout[site]):
                    if f"model.layers.{l}.{mod}" not in self.enabled:
                        continue
                    bound = 1.0 / math.sqrt(fin)
                    A[j * self.rp:j * self.rp + self.r, :fin] = (torch.rand((self.r, fin), generator=g) * 2 - 1) * bound
                    mA[j * self.rp:j * self.rp + self.r, :fin] = 1.0
                    mB[rows[:, None], torch.arange(j * self.rp, j * self.rp + self.r, device=dev)[None, :]] = 1.0
                self.master[(l, site, "A")] = A.to(dev)
                self.master[(l, site, "B")] = torch.zeros((n_out, rr), dtype=torch.float32, device=dev)
                self.mask[(l, site, "A")], self.mask[(l, site, "B")] = mA.to(dev), mB
                sizes += [(l, site, "A"), (l, site, "B")]
        total = sum(self.master[k].numel() for k in sizes)
        self.arena = torch.zeros(total, dtype=torch.float32, device=dev)
        at = 0
        for k in sizes:
            n = self.master[k].numel()
            self.grad[k] = self.arena[at:at + n].view(self.master[k].shape)
            self.m[k], self.v[k] = torch.zeros_like(self.master[k]), torch.zeros_like(self.master[k])
            at += n
        self.buckets = [("lora", 0, total)]
        self._finished = False
        self._fwd_id = 0
        for l in range(self.L):                          # gate/up: the projection with the adapter's K columns appended
            nt_k = text.shape[f"l{l}.gu_w"]
            base = text[f"l{l}.gu_w"]
            ks = nt_k[1] // 32
            rr = 2 * self.rp
            ext = torch.zeros((base.shape[0], (ks + rr // 32) * 512), dtype=torch.bfloat16, device=dev)
            ext.view(base.shape[0], ks + rr // 32, 512)[:, :ks] = base.view(base.shape[0], ks, 512)
            self.W.t[f"l{l}.gu_we"] = ext
            self.W.shape[f"l{l}.gu_we"] = (nt_k[0], nt_k[1] + rr)
        self.refresh()

    # ------------------------------------------------------------------ bookkeeping
    def module_names(self):
        return sorted(self.enabled)

    def n_params(self):
        return int(sum(m.sum().item() for m in self.mask.values()))

    def zero_grad(self):
        self.arena.zero_()
        self._finished = False

    def extra_k(self, site):
        return len(self.layout[site]) * self.rp

    def refresh(self):
        """fp32 masters -> the packed bf16 operands of the adapter GEMMs (after every optimiser step / load)"""
        s = self.scaling
        for l in range(self.L):
            for site in SITES:
                A = self.master[(l, site, "A")].to(torch.bfloat16)
                sB = (self.master[(l, site, "B")] * s).to(torch.bfloat16)
                n_out, rr = sB.shape
                pre = f"l{l}.{site}."
                self._set(pre + "A", A)                                            # t = x A^T            [N = r', K = in]
                self._set(pre + "At", A.t().contiguous())                          # g = dt A             [N = in, K = r']
                bt = torch.zeros((rr, pad64(n_out)), dtype=torch.bfloat16, device=A.device)
                bt[:, :n_out] = sB.t()
                self._set(pre + "Bt", bt)                                          # dt = dY (s B)        [N = r', K = out]
                if site == "gu":
                    ext = self.W.t[f"l{l}.gu_we"]
                    ks = self.text.shape[f"l{l}.gu_w"][1] // 32
                    ext.view(ext.shape[0], ks + rr // 32, 512)[:, ks:] = pack_tiles(sB).view(ext.shape[0], rr // 32, 512)
                else:
                    self._set(pre + "B", sB)                                       # y += t (s B)^T       [N = out, K = r']

    def _set(self, name, w):
        p = pack_tiles(w)
        if name in self.W.t and self.W.t[name].shape == p.shape:
            self.W.t[name].copy_(p)
        else:
            self.W.t[name] = p
        self.W.shape[name] = tuple(w.shape)

    # ------------------------------------------------------------------ forward / backward pieces (called by train.Trainer)
    def set_step(self, step):
        """forward ids restart from the optimiser step, so a resumed run redraws the masks the original run drew at that step"""
        self._fwd_id = int(step) << 20

    def begin_forward(self):
        self._fwd_id += 1
        return self._fwd_id

    def draw_keys(self, fid, l, site):
        """(seed, salt) of one site's dropout draw: the 32-bit seed mixes the run's seed with the forward's id, the salt names the site"""
        x = (self.seed * 0x9E3779B1 + (fid & 0xFFFFFFFF) * 0x85EBCA77 + (fid >> 32) * 0xC2B2AE3D + 0x27D4EB2F) & 0xFFFFFFFF
        x ^= x >> 15
        x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
        x ^= x >> 12
        return x, l * 8 + SITES.index(site)

    def _dropped(self, x, T, K, keys, tag):
        if self.p == 0.0:
            return x
        xd = self.eng.buf("lora.xd." + tag, (T, K), torch.bfloat16)
        lib.call("vr2_dropout_bf16", x, x.stride(0), xd, xd.stride(0), T, K, self.p, keys[0], keys[1], self.eng.s)
        return xd

    def project(self, fid, l, site, x, t_out, T):
        """t_out[T, r'] (bf16, may be a strided view) = dropout(x) A^T"""
        kin = self.W.shape[f"l{l}.{site}.A"][1]
        xd = self._dropped(x, T, kin, self.draw_keys(fid, l, site), "f")
        self.eng._gemm(xd, self.W, f"l{l}.{site}.A", None, t_out, T, EPI_BF16)

    def join(self, l, site, t, out_f32, T):
        """out_f32 += t (s B)^T  (qkv / o / down; gate/up joins inside its own projection)"""
        self.eng._gemm(t, self.W, f"l{l}.{site}.B", None, out_f32, T, EPI_ADD_F32)

    def backward(self, fid, l, site, dY, x, t, dst, T):
        """the adapter's share of one site's backward: dA += dt^T dropout(x), dB += dY^T t (scaled by s in finish_grads),
        dst (the frozen base layer's input gradient, fp32 or bf16) += mask o (dt A) / (1 - p)"""
        eng = self.eng
        rr = self.extra_k(site)
        n_out = self.master[(l, site, "B")].shape[0]
        kin = self.W.shape[f"l{l}.{site}.A"][1]
        dt = eng.buf("lora.dt", (T, 3 * self.rp), torch.bfloat16)[:, :rr]
        eng._gemm(dY, self.W, f"l{l}.{site}.Bt", None, dt, T, EPI_BF16)
        gB, gA = self.grad[(l, site, "B")], self.grad[(l, site, "A")]
        lib.call("vr2_gemm_tn_bf16", dY, dY.stride(0), t, t.stride(0), gB, gB.stride(0), n_out, rr, T, 1, eng.s)
        keys = self.draw_keys(fid, l, site)
        xd = self._dropped(x, T, kin, keys, "b")
        lib.call("vr2_gemm_tn_bf16", dt, dt.stride(0), xd, xd.stride(0), gA, gA.stride(0), rr, kin, T, 1, eng.s)
        g32 = eng.buf("lora.g32", (T, kin), torch.float32)
        eng._gemm(dt, self.W, f"l{l}.{site}.At", None, g32, T, EPI_F32)
        lib.call("vr2_dropout_add", g32, g32.stride(0), dst, dst.stride(0), 1 if dst.dtype == torch.bfloat16 else 0, T, kin, self.p,
                 keys[0], keys[1], eng.s)

    def merge_into(self, dst: Params, base: Params):
        """dst's projections <- base W + s B A (fp32 sum, one rounding, re-packed): the adapted model as plain weights, for the passes
        that take no gradient (rollout prefill / decode, old-policy scoring) -- peft's merge(), without touching the base copies the
        training forward and the adapter-disabled KL reference read"""
        from .weights import unpack_tiles
        with torch.cuda.stream(self.eng.stream):
            for l in range(self.L):
                for site in SITES:
                    name = f"l{l}.{self.base[site]}"
                    n, kp = base.shape[name]
                    w = unpack_tiles(base[name], n, kp).float()
                    w.addmm_(self.master[(l, site, "B")], self.master[(l, site, "A")], alpha=self.scaling)
                    dst.t[name].copy_(pack_tiles(w.to(torch.bfloat16)))

    def finish_grads(self):
        """once per optimiser step, before the exchange / the norm: the B gradients take their factor s, and everything outside the
        adapters' own blocks (the fused GEMMs fill whole rectangles) is cleared"""
        if self._finished:
            return
        for k, g in self.grad.items():
            g.mul_(self.mask[k] * (self.scaling if k[2] == "B" else 1.0))
        self._finished = True

    # ------------------------------------------------------------------ peft layout
    def adapter_state_dict(self):
        out = {}
        for l in range(self.L):
            for site in SITES:
                A, B = self.master[(l, site, "A")], self.master[(l, site, "B")]
                for j, (mod, fin, rows) in enumerate(self.layout[site]):
                    if f"model.layers.{l}.{mod}" not in self.enabled:
                        continue
                    pre = f"base_model.model.model.layers.{l}.{mod}."
                    out[pre + "lora_A.weight"] = A[j * self.rp:j * self.rp + self.r, :fin].contiguous().cpu()
                    out[pre + "lora_B.weight"] = B[rows][:, j * self.rp:j * self.rp + self.r].contiguous().cpu()
        return out

    def load_adapter_state_dict(self, sd):
        for l in range(self.L):
            for site in SITES:
                A, B = self.master[(l, site, "A")], self.master[(l, site, "B")]
                for j, (mod, fin, rows) in enumerate(self.layout[site]):
                    pre = f"base_model.model.model.layers.{l}.{mod}."
                    if pre + "lora_A.weight" not in sd:
                        continue
                    A[j * self.rp:j * self.rp + self.r, :fin] = sd[pre + "lora_A.weight"].to(A)
                    cols = torch.arange(j * self.rp, j * self.rp + self.r, device=B.device)
                    B[rows[:, None], cols[None, :]] = sd[pre + "lora_B.weight"].to(B)
        self.refresh()

    def save_adapter(self, out_dir, base_model_name_or_path=""):
        from safetensors.torch import save_file
        os.makedirs(out_dir, exist_ok=True)
        save_file(self.adapter_state_dict(), os.path.join(out_dir, "adapter_model.safetensors"), metadata={"format": "pt"})
        cfg = {"peft_type": "LORA", "task_type": None, "base_model_name_or_path": base_model_name_or_path, "r": self.r,
               "lora_alpha": self.alpha if self.alpha != int(self.alpha) else int(self.alpha), "lora_dropout": self.p, "bias": "none",
               "target_modules": self.module_names(), "fan_in_fan_out": False, "inference_mode": True, "init_lora_weights": True,
               "modules_to_save": None, "use_dora": False, "use_rslora": False}
        with open(os.path.join(out_dir, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2)

    def save_training_state(self, ckpt_dir):
        st = {"master": {"|".join(map(str, k)): v.cpu() for k, v in self.master.items()},
              "m": {"|".join(map(str, k)): v.cpu() for k, v in self.m.items()}, "v": {"|".join(map(str, k)): v.cpu() for k, v in self.v.items()}}
        torch.save(st, os.path.join(ckpt_dir, "lora_state.pt"))

    def load_training_state(self, ckpt_dir):
        st = torch.load(os.path.join(ckpt_dir, "lora_state.pt"), map_location="cpu", weights_only=True)
        for k in self.master:
            key = "|".join(map(str, k))
            self.master[k].copy_(st["master"][key]); self.m[k].copy_(st["m"][key]); self.v[k].copy_(st["v"][key])
        self.refresh()


def merged_state_dict(base_sd, adapter_sd, scaling):
    """merge_and_unload(): W + scaling * lora_B @ lora_A on every adapted linear layer of an HF state dict (fp32 sum, one rounding to
    the base tensor's dtype).  base keys: the legacy / 5.x layouts weights.normalise_keys understands."""
    from .weights import normalise_keys
    sd = dict(normalise_keys(base_sd))
    for k in [k for k in adapter_sd if k.endswith("lora_A.weight")]:
        mod = k[len("base_model.model."):-len(".lora_A.weight")]
        A, B = adapter_sd[k].float(), adapter_sd[k.replace("lora_A", "lora_B")].float()
        name = mod + ".weight"
        sd[name] = (sd[name].float() + scaling * (B @ A)).to(sd[name].dtype)
    return sd
