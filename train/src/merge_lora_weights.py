"""Fold LoRA adapters into their base model (same path and flags as the reference's train/src/merge_lora_weights.py:4-21:
`load_pretrained_model(model_path, model_base)` -> PeftModel.merge_and_unload() -> save_pretrained).

    python src/merge_lora_weights.py --model-path <adapter dir> --model-base <HF base dir> --save-model-path <out> [--safe-serialization]

<adapter dir> holds adapter_model.safetensors + adapter_config.json (and non_lora_state_dict.bin, the trainable non-adapter tensors of
the SFT run, /root/reference/train/src/utils.py:66-72: loaded over the base weights first, strict=False).  The output is an HF
Qwen2_5_VLForConditionalGeneration directory (safetensors + config.json + the base's tokenizer / processor files): what the GRPO stage
takes as --model_id.  Host-side tensor arithmetic only (W + alpha / r * B @ A in fp32, one rounding): no GPU needed.
"""
import argparse
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
for p_ in (ROOT, os.path.join(os.path.dirname(os.path.abspath(__file__)), "train")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def merge_lora(args):
    import torch
    from safetensors.torch import save_file
    from video_r2_amd import weights as W
    from train_grpo import copy_processor_files
    cfg = json.load(open(os.path.join(args.model_path, "adapter_config.json")))
    if cfg.get("peft_type") != "LORA" or cfg.get("use_dora") or cfg.get("bias", "none") != "none":
        raise NotImplementedError(f"adapter_config.json: {cfg.get('peft_type')} / use_dora={cfg.get('use_dora')} / bias={cfg.get('bias')}: plain LoRA with bias none is what merges here")
    merged = W.load_state_dict(args.model_path, model_base=args.model_base)     # base + non-LoRA trainables + W + alpha / r * B A
    os.makedirs(args.save_model_path, exist_ok=True)
    src_cfg = os.path.join(args.model_path if os.path.exists(os.path.join(args.model_path, "config.json")) else args.model_base, "config.json")
    with open(src_cfg) as f, open(os.path.join(args.save_model_path, "config.json"), "w") as g:
        g.write(f.read())
    tensors = {k: v.contiguous() for k, v in merged.items()}
    if args.safe_serialization:
        save_file(tensors, os.path.join(args.save_model_path, "model.safetensors"), metadata={"format": "pt"})
    else:
        torch.save(tensors, os.path.join(args.save_model_path, "pytorch_model.bin"))
    copy_processor_files(args.model_base, args.save_model_path)
    return args.save_model_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", type=str, required=True)
    ap.add_argument("--model-base", type=str, required=True)
    ap.add_argument("--save-model-path", type=str, required=True)
    ap.add_argument("--safe-serialization", action="store_true")
    print(merge_lora(ap.parse_args()))
