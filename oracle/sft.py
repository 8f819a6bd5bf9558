"""Stage-1 SFT of the reference, restated on the CPU oracle: the supervised loss and the LoRA branch.

Test infrastructure only (see oracle/__init__.py).

* `sft_loss`: the loss of the mixed-modality forward (/root/reference/train/src/train/monkey_patch_forward.py:619-631): logits to
  fp32, shift so that position t predicts token t + 1, `CrossEntropyLoss()` (mean over the positions whose label is not -100) over
  the flattened batch.  PINNED: tests/golden/sft_tiny_g2.npz (oracle/make_golden.py case_sft) holds HF transformers' own loss for
  `labels` and HF autograd's gradients of six tensors (LM, lm_head, merger, ViT block) on a video row with two supervised spans; this
  function over oracle forward_logits reproduces the loss exactly and the gradients to 1e-6 (tests/test_sft_cpu.py).
* `labels_of`: the per-piece label rule of the dataset (/root/reference/train/src/dataset/sft_dataset.py:127-176): system and
  user pieces (up to and including the answer's role header) are -100, the answer piece (`{answer}<|im_end|>\\n`) is supervised.
* `LoraOracle`: the adapters `get_peft_model(model, LoraConfig(r, lora_alpha, target_modules, lora_dropout, bias="none"))` puts on
  the LM's linear layers (train_sft.py:153-169; target names from find_target_linear_names :24-39).  peft is not importable here
  and the reference pins no peft version; what is restated is peft's published `lora.Linear.forward`,
      result = base_layer(x) + lora_B(lora_A(lora_dropout(x))) * scaling,    scaling = lora_alpha / r,
  with lora_A initialised kaiming-uniform(a = sqrt(5)) (= U(-1/sqrt(in), 1/sqrt(in))) and lora_B zero, and `merge`'s
  W + scaling * B @ A.  No reference-held fixture covers it: "parity unpinned" beyond that published formula.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100


def sft_loss(logits, labels):
    lg = logits.float()[..., :-1, :].contiguous()
    lb = torch.as_tensor(labels)[..., 1:].contiguous()
    return F.cross_entropy(lg.view(-1, lg.shape[-1]), lb.view(-1), ignore_index=IGNORE_INDEX)


def labels_of(pieces):
    """pieces: [(ids, supervised)] in sequence order -> (input_ids, labels)"""
    ids, lab = [], []
    for p, sup in pieces:
        ids.extend(p)
        lab.extend(p if sup else [IGNORE_INDEX] * len(p))
    return np.asarray(ids, dtype=np.int64), np.asarray(lab, dtype=np.int64)


LM_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


def target_module_names(n_layers, num_lora_modules=-1):
    """find_target_linear_names over the LM with lm_head / embed_tokens / visual excluded: module order of the HF tree, the last
    `num_lora_modules` of them when that is positive"""
    names = [f"model.layers.{i}.{t}" for i in range(n_layers) for t in LM_TARGETS]
    return names[-num_lora_modules:] if num_lora_modules > 0 else names


class LoraOracle:
    def __init__(self, w, n_layers, r, alpha, seed=0, num_lora_modules=-1, b_std=0.0):
        """A ~ U(-1/sqrt(in), 1/sqrt(in)), B = 0 (or N(0, b_std) so that tests see a non-trivial branch)"""
        g = torch.Generator().manual_seed(seed)
        self.scaling = alpha / r
        self.A, self.B, self.masks, self.keep = {}, {}, {}, 1.0
        for m in target_module_names(n_layers, num_lora_modules):
            out_f, in_f = w[m + ".weight"].shape
            bound = 1.0 / math.sqrt(in_f)
            self.A[m] = ((torch.rand(r, in_f, generator=g) * 2 - 1) * bound).requires_grad_(True)
            self.B[m] = (b_std * torch.randn(out_f, r, generator=g)).requires_grad_(True)

    def set_dropout(self, masks, p):
        """masks: {module name: 0/1 tensor broadcastable to the module's input} (the engine's own draws, so that both sides drop
        the same elements); kept elements are scaled by 1 / (1 - p) as torch.nn.Dropout does"""
        self.masks, self.keep = masks, 1.0 - p

    def hooks(self):
        def branch(m):
            def f(x):
                if m in self.masks:
                    x = x * self.masks[m].to(x.dtype) / self.keep
                return (x @ self.A[m].t()) @ self.B[m].t() * self.scaling
            return f
        return {m + ".weight": branch(m) for m in self.A}

    def merged(self, w):
        """merge_and_unload(): W + scaling * B @ A on every adapted layer"""
        out = dict(w)
        for m in self.A:
            out[m + ".weight"] = w[m + ".weight"] + self.scaling * (self.B[m] @ self.A[m]).detach()
        return out
