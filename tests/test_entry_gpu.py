"""GPU: the shell entry point trains.  `bash train/scripts/train_grpo.sh` (the reference's launch line: torchrun ->
src/train/train_grpo.py, /root/reference/train/scripts/train_grpo.sh:41-80) on a tiny HF-layout model directory, a 6-row JSON
dataset of array clips with one subtitle file, overlays on: three optimiser steps with checkpoints every two, then a second
launch that must resume from checkpoint-2 (train_grpo.py:184-187) instead of starting over."""
import ast
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import model as om
from tests.helpers import to_engine_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(env, *flags):
    r = subprocess.run(["bash", os.path.join(ROOT, "train", "scripts", "train_grpo.sh"), *flags], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n---- stderr ----\n" + r.stderr[-6000:]
    steps = [ast.literal_eval(ln) for ln in r.stdout.splitlines() if ln.startswith("{'step'")]
    return steps, r.stderr


def test_train_grpo_sh_trains_checkpoints_and_resumes(tmp_path):
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=41), device="cuda:0")
    model_dir = tmp_path / "sft_model"
    m.save_pretrained(str(model_dir))
    (model_dir / "generation_config.json").write_text("{}")               # a processor-side file that must travel with the weights
    del m
    clips, subs = tmp_path / "videos", tmp_path / "subs"
    (clips / "set").mkdir(parents=True)
    (subs / "set").mkdir(parents=True)
    rng = np.random.default_rng(0)
    rows = []
    for i in range(6):
        np.savez(clips / "set" / f"c{i}.npz", frames=rng.integers(0, 256, size=(60, 56, 84, 3), dtype=np.uint8), fps=np.float32(30.0))
        rows.append({"video": f"set/c{i}.npz", "key_frame": "",
                     "conversations": [{"from": "human", "value": f"<video>\nQuestion {i}?\nA. yes\nB. no"},
                                       {"from": "gpt", "value": "<think>..</think><answer>B</answer>"}],
                     "temporal_grounding": {"00:00-00:01": "something happens."}})
    (subs / "set" / "c0.srt").write_text("1\n00:00:00,000 --> 00:00:02,000\nhello world\n")
    data = tmp_path / "grpo.json"
    data.write_text(json.dumps(rows))
    # the user's reward module (the --reward_module seam): the built-in rewards plus one that varies inside a group even for a
    # random-init model, so that advantages -- and gradients -- are not all zero
    (tmp_path / "my_reward_funcs.py").write_text(
        "from video_r2_amd.rewards import accuracy_reward, format_reward  # noqa: F401\n\n\n"
        "def text_length_reward(completions, assistant, temporal_grounding, **kwargs):\n"
        "    assert all(isinstance(a, dict) and 'content' in a for a in assistant) and len(temporal_grounding) == len(completions)\n"
        "    assert kwargs['prompts'][0][1]['content'][1]['text'].startswith('Question')\n"
        "    return [float(i % 2) + 0.1 * (len(c[0]['content']) % 3) for i, c in enumerate(completions)]\n")
    out_dir = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}:{os.environ.get('PYTHONPATH', '')}", NUM_DEVICES="1", MODEL_NAME=str(model_dir), DATA_PATH=str(data), DATA_FOLDER=str(clips),
               SUBTITLES_FOLDER=str(subs), OUTPUT_DIR=str(out_dir), FPS_MAX_FRAMES="4", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    flags = ["--num_generations", "4", "--per_device_train_batch_size", "4", "--max_completion_length", "12",
             "--reward_module", "my_reward_funcs", "--reward_func_names", "accuracy_reward,format_reward,text_length_reward",
             "--reward_weights", "1.0,1.0,0.5", "--logging_steps", "1",
             "--save_steps", "2", "--dataloader_num_workers", "2", "--learning_rate", "1e-3", "--video_min_pixels", "4704",
             "--video_max_pixels", "4704", "--log_completions", "True"]
    steps, err = _launch(env, *flags, "--max_steps", "3")
    assert [s["step"] for s in steps] == [1, 2, 3], err[-3000:]
    for s in steps:
        assert np.isfinite(s["loss"]) and s["grad_norm"] > 0 and s["completions/mean_length"] > 0
        for k in ("reward", "reward_std", "rewards/accuracy_reward/mean", "rewards/accuracy_reward/std", "rewards/text_length_reward/std",
                  "kl", "clip_ratio/region_mean", "learning_rate", "num_tokens"):
            assert k in s, k
    assert "ByteTokenizer" in err and "reward functions ['accuracy_reward', 'format_reward', 'text_length_reward']" in err and "from my_reward_funcs" in err
    assert sorted(p for p in os.listdir(out_dir) if p.startswith("checkpoint-")) == ["checkpoint-2"]
    for d in (out_dir, out_dir / "checkpoint-2"):
        assert (d / "model.safetensors").exists() and (d / "config.json").exists() and (d / "generation_config.json").exists()
    assert (out_dir / "checkpoint-2" / "trainer_state.pt").exists()
    # HF Trainer's bookkeeping files: the log history in trainer_state.json (final directory and checkpoints), the metrics stream
    st = json.loads((out_dir / "trainer_state.json").read_text())
    assert st["global_step"] == 3 and [r["step"] for r in st["log_history"]] == [1, 2, 3] and "reward" in st["log_history"][0]
    assert json.loads((out_dir / "checkpoint-2" / "trainer_state.json").read_text())["global_step"] == 2
    assert len((out_dir / "runs" / "metrics.jsonl").read_text().splitlines()) == 3
    # --log_completions: the reference's completions table (grpo_trainer.py:1206-1227: step, prompt, completion, one column per reward
    # function) as rows of runs/completions.jsonl -- 3 steps x 1 prompt group x 4 completions
    table = [json.loads(ln) for ln in (out_dir / "runs" / "completions.jsonl").read_text().splitlines()]
    assert len(table) == 12 and {r["step"] for r in table} == {0, 1, 2}
    assert all({"prompt", "completion", "reward", "accuracy_reward", "format_reward", "text_length_reward"} <= set(r) for r in table)
    # second launch, one more step allowed: resumes after step 2 (steps 3 and 4 run, not 1..4) and keeps --save_total_limit 2
    steps2, err2 = _launch(env, *flags, "--max_steps", "4")
    assert "resumed from" in err2 and "checkpoint-2 at step 2" in err2
    assert [s["step"] for s in steps2] == [3, 4]
    assert [r["step"] for r in json.loads((out_dir / "trainer_state.json").read_text())["log_history"]] == [1, 2, 3, 4]   # history continued
    assert sorted(p for p in os.listdir(out_dir) if p.startswith("checkpoint-")) == ["checkpoint-2", "checkpoint-4"]
    # the resumed step 3 sees the same prompts and the same sampling seed as the first run's step 3, from the same weights
    assert steps2[0]["completions/mean_length"] == steps[2]["completions/mean_length"]
    assert steps2[0]["loss"] == pytest.approx(steps[2]["loss"], rel=1e-5, abs=1e-7)
    # third launch into a fresh output directory: BASELINE config 5 (fp8 decode + fp8 no-grad GEMMs) with HF's top-p / min-p /
    # repetition-penalty flags and the phase timers, through the same shell entry -- the flags must reach the engine, not be dropped
    env3 = dict(env, OUTPUT_DIR=str(tmp_path / "out_fp8"))
    steps3, err3 = _launch(env3, *flags, "--max_steps", "2", "--decode_weights", "fp8", "--top_p", "0.9", "--min_p", "0.02",
                           "--repetition_penalty", "1.1", "--top_k", "40")
    assert [s["step"] for s in steps3] == [1, 2], err3[-3000:]
    for s in steps3:
        assert np.isfinite(s["loss"]) and s["grad_norm"] > 0
        assert s["profiling/Time taken: QwenGRPOTrainer.generate"] > 0 and "profiling/Time taken: QwenGRPOTrainer.text_length_reward" in s
    # fourth launch: dataset rows with an "image" key (grpo_dataset.py:113-125) through the same shell entry -- fetch_image's
    # stages on the host, image-pad tokens, the one-frame vision items beside nothing else in the tower's pass
    from PIL import Image
    img_rows = []
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, size=(60 + 10 * i, 90, 3), dtype=np.uint8)).save(clips / "set" / f"p{i}.png")
        img_rows.append({"image": f"set/p{i}.png",
                         "conversations": [{"from": "human", "value": f"<image>\nQuestion {i} about the picture?\nA. yes\nB. no"},
                                           {"from": "gpt", "value": "<answer>A</answer>"}], "temporal_grounding": {}})
    data_img = tmp_path / "grpo_img.json"
    data_img.write_text(json.dumps(img_rows))
    env4 = dict(env, OUTPUT_DIR=str(tmp_path / "out_img"), DATA_PATH=str(data_img))
    steps4, err4 = _launch(env4, *flags, "--max_steps", "2", "--image_min_pixels", "3136", "--image_max_pixels", "4704")
    assert [s["step"] for s in steps4] == [1, 2], err4[-3000:]
    for s in steps4:
        assert np.isfinite(s["loss"]) and s["grad_norm"] > 0 and s["completions/mean_length"] > 0
    # fifth launch: the vision side trainable (train_grpo.py:48-57 of the reference) through the same shell entry -- the tower and the
    # merger of the saved model moved away from the SFT model's, the LM's too
    from safetensors.torch import load_file
    env5 = dict(env, OUTPUT_DIR=str(tmp_path / "out_vis"))
    steps5, err5 = _launch(env5, *flags, "--max_steps", "2", "--freeze_vision_tower", "False", "--freeze_merger", "False")
    assert [s["step"] for s in steps5] == [1, 2], err5[-3000:]
    assert steps5[0]["kl"] == 0.0 and steps5[1]["kl"] > 0.0 and all(np.isfinite(s["loss"]) and s["grad_norm"] > 0 for s in steps5)
    assert "vision tower" in err5 and "merger" in err5
    sft, got = load_file(str(model_dir / "model.safetensors")), load_file(str(tmp_path / "out_vis" / "model.safetensors"))
    for k in ("visual.blocks.0.attn.qkv.weight", "visual.blocks.3.mlp.down_proj.bias", "visual.patch_embed.proj.weight",
              "visual.merger.mlp.2.weight", "visual.merger.mlp.0.bias", "model.layers.0.self_attn.o_proj.weight"):   # (norm weights near 1
        # move by less than half a bf16 ulp of the saved file in two steps of 1e-3; their fp32 masters are in trainer_state.pt)
        assert not np.array_equal(sft[k].float().numpy(), got[k].float().numpy()), k
    # sixth launch: --freeze_llm True under a trainable merger (configure_llm + configure_vision_tower): the merger moves, the LM and the
    # frozen tower are written back exactly as they came
    env6 = dict(env, OUTPUT_DIR=str(tmp_path / "out_merger"))
    steps6, err6 = _launch(env6, *flags, "--max_steps", "2", "--freeze_llm", "True", "--freeze_merger", "False")
    assert [s["step"] for s in steps6] == [1, 2], err6[-3000:]
    assert all(np.isfinite(s["loss"]) and s["grad_norm"] > 0 for s in steps6)
    got6 = load_file(str(tmp_path / "out_merger" / "model.safetensors"))
    assert not np.array_equal(sft["visual.merger.mlp.2.weight"].float().numpy(), got6["visual.merger.mlp.2.weight"].float().numpy())
    for k in ("visual.blocks.0.attn.qkv.weight", "model.layers.0.self_attn.o_proj.weight", "lm_head.weight", "model.embed_tokens.weight"):
        assert np.array_equal(sft[k].float().numpy(), got6[k].float().numpy()), k


def test_train_sft_sh_lora_recipe_then_merge_then_full_fine_tune(tmp_path):
    """`bash train/scripts/train_sft.sh` (the reference's launch line, scripts/train_sft.sh:27-66: LoRA on the LM's linear layers,
    everything else frozen) on a tiny HF-layout model directory and a 6-row dataset (array clips, one still image, one text-only row, a
    two-answer conversation): adapters in peft's layout + non_lora_state_dict.bin; `merge_lora_weights.py` folds them into an HF
    directory that loads; a second launch with --lora_enable False trains the LM and the merger, checkpoints and resumes."""
    import torch
    from PIL import Image
    from safetensors.torch import load_file
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=43), device="cuda:0")
    model_dir = tmp_path / "base_model"
    m.save_pretrained(str(model_dir))
    (model_dir / "generation_config.json").write_text("{}")
    del m
    clips = tmp_path / "videos"
    (clips / "set").mkdir(parents=True)
    rng = np.random.default_rng(1)
    rows = []
    for i in range(4):
        np.savez(clips / "set" / f"c{i}.npz", frames=rng.integers(0, 256, size=(60, 56, 84, 3), dtype=np.uint8), fps=np.float32(30.0))
        conv = [{"from": "human", "value": f"<video>\nWhat happens in clip {i}?"}, {"from": "gpt", "value": f"<think>00:01 something</think><answer>{'AB'[i % 2]}</answer>"}]
        if i == 0:
            conv += [{"from": "human", "value": "And after that?"}, {"from": "gpt", "value": "Nothing else."}]
        rows.append({"video": f"set/c{i}.npz", "conversations": conv})
    Image.fromarray(rng.integers(0, 256, size=(60, 90, 3), dtype=np.uint8)).save(clips / "set" / "p.png")
    rows.append({"image": "set/p.png", "conversations": [{"from": "human", "value": "<image>\nWhat is shown?"}, {"from": "gpt", "value": "A picture."}]})
    rows.append({"conversations": [{"from": "human", "value": "Say hello."}, {"from": "gpt", "value": "Hello."}]})
    data = tmp_path / "sft.json"
    data.write_text(json.dumps(rows))
    out_dir = tmp_path / "out_lora"
    env = dict(os.environ, NUM_DEVICES="1", GLOBAL_BATCH_SIZE="2", BATCH_PER_DEVICE="1", MODEL_NAME=str(model_dir), DATA_PATH=str(data), DATA_FOLDER=str(clips),
               SUBTITLES_FOLDER=str(tmp_path / "subs"), OUTPUT_DIR=str(out_dir), FPS_MAX_FRAMES="4", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)

    def launch(env_, *flags):
        r = subprocess.run(["bash", os.path.join(ROOT, "train", "scripts", "train_sft.sh"), *flags], env=env_, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + "\n---- stderr ----\n" + r.stderr[-6000:]
        return [ast.literal_eval(ln) for ln in r.stdout.splitlines() if ln.startswith("{'step'")], r.stderr

    small = ["--video_min_pixels", "4704", "--video_max_pixels", "4704", "--image_min_pixels", "3136", "--image_max_pixels", "4704", "--lora_rank", "8",
             "--lora_alpha", "16", "--learning_rate", "1e-3", "--dataloader_num_workers", "2"]
    steps, err = launch(env, *small)
    assert [s["step"] for s in steps] == [1, 2, 3], err[-3000:]                       # 6 rows / (1 x 2 accumulated x 1 rank), one epoch
    assert all(np.isfinite(s["loss"]) and s["loss"] > 0 for s in steps) and steps[0]["learning_rate"] == 0.0 and steps[1]["grad_norm"] > 0
    assert "Adding LoRA to the model... 14 lora modules" in err and "Trainable parameters:" in err and "1 warm-up step" in err
    ad = load_file(str(out_dir / "adapter_model.safetensors"))
    assert len(ad) == 28 and ad["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"].shape == (8, 512)
    assert float(ad["base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"].abs().max()) > 0           # B left zero: the adapters trained
    cfgj = json.loads((out_dir / "adapter_config.json").read_text())
    assert cfgj["r"] == 8 and cfgj["lora_alpha"] == 16 and cfgj["lora_dropout"] == 0.05 and cfgj["base_model_name_or_path"] == str(model_dir)
    assert torch.load(str(out_dir / "non_lora_state_dict.bin"), weights_only=True) == {} and (out_dir / "config.json").exists()
    assert not (out_dir / "model.safetensors").exists()                                  # adapters only, as the reference saves them
    # merge_lora_weights.py: adapters folded into the base -> an HF directory the engine (and HF) loads
    merged_dir = tmp_path / "merged"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train", "src", "merge_lora_weights.py"), "--model-path", str(out_dir), "--model-base", str(model_dir),
                        "--save-model-path", str(merged_dir), "--safe-serialization"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    base_sd, merged_sd = load_file(str(model_dir / "model.safetensors")), load_file(str(merged_dir / "model.safetensors"))
    k = "model.layers.1.mlp.down_proj.weight"
    want = base_sd[k].float() + 2.0 * ad["base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"].float() @ ad["base_model.model.model.layers.1.mlp.down_proj.lora_A.weight"].float()
    assert torch.equal(merged_sd[k], want.to(torch.bfloat16)) and not torch.equal(merged_sd[k], base_sd[k])
    assert torch.equal(merged_sd["visual.merger.mlp.2.weight"], base_sd["visual.merger.mlp.2.weight"]) and (merged_dir / "generation_config.json").exists()
    m_merged = VideoR2Model.from_pretrained(str(merged_dir))
    m_direct = VideoR2Model.from_pretrained(str(out_dir))             # the adapter directory itself: merged on load over base_model_name_or_path
    assert torch.equal(m_direct.text["l1.down_w"], m_merged.text["l1.down_w"]) and torch.equal(m_direct.text["l0.qkv_w"], m_merged.text["l0.qkv_w"])
    del m_merged, m_direct
    # full fine-tune of the LM + the merger (no adapters), checkpoints every step, then a resumed launch
    env2 = dict(env, OUTPUT_DIR=str(tmp_path / "out_full"))
    full = small + ["--lora_enable", "False", "--freeze_llm", "False", "--freeze_merger", "False", "--save_strategy", "steps", "--save_steps", "1",
                    "--merger_lr", "5e-4"]
    steps2, err2 = launch(env2, *full, "--max_steps", "2")
    assert [s["step"] for s in steps2] == [1, 2], err2[-3000:]
    got = load_file(str(tmp_path / "out_full" / "model.safetensors"))
    assert not torch.equal(got["model.layers.0.self_attn.o_proj.weight"], base_sd["model.layers.0.self_attn.o_proj.weight"])
    assert not torch.equal(got["visual.merger.mlp.2.weight"], base_sd["visual.merger.mlp.2.weight"])
    assert torch.equal(got["visual.blocks.0.attn.qkv.weight"], base_sd["visual.blocks.0.attn.qkv.weight"])            # the tower stayed frozen
    steps3, err3 = launch(env2, *full, "--max_steps", "3")
    assert "resumed from" in err3 and [s["step"] for s in steps3] == [3]
    # two ranks through the same shell entry (both on this GPU, gloo collectives): rows sharded batch by batch, the adapters' gradient
    # bucket averaged, rank 0 writes the adapters
    env3 = dict(env, OUTPUT_DIR=str(tmp_path / "out_lora_2ranks"), NUM_DEVICES="2", GLOBAL_BATCH_SIZE="2", VR2_DIST_BACKEND="gloo", MASTER_PORT="29533")
    steps4, err4 = launch(env3, *small, "--max_steps", "2")
    assert [s["step"] for s in steps4] == [1, 2], err4[-3000:]
    assert "x 2 rank(s)" in err4 and all(np.isfinite(s["loss"]) for s in steps4) and steps4[1]["grad_norm"] > 0
    ad2 = load_file(str(tmp_path / "out_lora_2ranks" / "adapter_model.safetensors"))
    assert len(ad2) == 28 and float(ad2["base_model.model.model.layers.0.mlp.gate_proj.lora_B.weight"].abs().max()) > 0


def test_train_grpo_sh_with_lora_adapters(tmp_path):
    """the GRPO entry with --lora_enable True (train_grpo.py:137-149,193-204 of the reference): the policy is base + adapters, the KL
    reference the adapter-disabled model (kl == 0 at step 1 where B = 0, > 0 after the first update), rollouts come from the merged copy,
    the output is the adapters in peft's layout and the base weights are written back untouched"""
    import torch
    from safetensors.torch import load_file
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    m = VideoR2Model.from_state_dict(to_engine_cfg(ocfg), om.make_weights(ocfg, seed=44), device="cuda:0")
    model_dir = tmp_path / "sft_model"
    m.save_pretrained(str(model_dir))
    base_sd = {k: v.clone() for k, v in m.state_dict().items()}
    del m
    clips = tmp_path / "videos"
    (clips / "set").mkdir(parents=True)
    rng = np.random.default_rng(2)
    rows = []
    for i in range(4):
        np.savez(clips / "set" / f"c{i}.npz", frames=rng.integers(0, 256, size=(60, 56, 84, 3), dtype=np.uint8), fps=np.float32(30.0))
        rows.append({"video": f"set/c{i}.npz", "key_frame": "",
                     "conversations": [{"from": "human", "value": f"<video>\nQuestion {i}?\nA. yes\nB. no"}, {"from": "gpt", "value": "<answer>B</answer>"}],
                     "temporal_grounding": {}})
    data = tmp_path / "grpo.json"
    data.write_text(json.dumps(rows))
    (tmp_path / "my_reward_funcs.py").write_text(
        "def text_length_reward(completions, **kwargs):\n"
        "    return [float(i % 2) + 0.1 * (len(c[0]['content']) % 3) for i, c in enumerate(completions)]\n")
    out_dir = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}:{os.environ.get('PYTHONPATH', '')}", NUM_DEVICES="1", MODEL_NAME=str(model_dir), DATA_PATH=str(data),
               DATA_FOLDER=str(clips), SUBTITLES_FOLDER=str(tmp_path / "subs"), OUTPUT_DIR=str(out_dir), FPS_MAX_FRAMES="4", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    flags = ["--num_generations", "4", "--per_device_train_batch_size", "4", "--max_completion_length", "10", "--reward_module", "my_reward_funcs",
             "--reward_func_names", "text_length_reward", "--reward_weights", "1.0", "--logging_steps", "1", "--save_steps", "2", "--learning_rate", "2e-3",
             "--video_min_pixels", "4704", "--video_max_pixels", "4704", "--freeze_llm", "True", "--lora_enable", "True", "--lora_rank", "8", "--lora_alpha", "16",
             "--lora_dropout", "0.05", "--lora_namespan_exclude", "['lm_head', 'embed_tokens']"]
    steps, err = _launch(env, *flags, "--max_steps", "3")
    assert [s["step"] for s in steps] == [1, 2, 3], err[-3000:]
    assert steps[0]["kl"] == 0.0 and steps[2]["kl"] > 0.0 and all(np.isfinite(s["loss"]) and s["grad_norm"] > 0 for s in steps)
    assert "LoRA: 14 adapted modules" in err
    ad = load_file(str(out_dir / "adapter_model.safetensors"))
    assert len(ad) == 28 and float(ad["base_model.model.model.layers.0.self_attn.v_proj.lora_B.weight"].abs().max()) > 0
    non = torch.load(str(out_dir / "non_lora_state_dict.bin"), weights_only=True)
    assert all(torch.equal(non["base_model.model." + k], v.cpu()) for k, v in base_sd.items())          # the base model did not move
    ck = out_dir / "checkpoint-2"
    assert (ck / "adapter_model.safetensors").exists() and (ck / "lora_state.pt").exists() and not (ck / "model.safetensors").exists()
    steps2, err2 = _launch(env, *flags, "--max_steps", "4")                                            # resumes from the adapters of step 2
    assert "resumed from" in err2 and [s["step"] for s in steps2] == [3, 4]
    assert steps2[0]["loss"] == pytest.approx(steps[2]["loss"], rel=1e-4, abs=1e-6) and steps2[0]["kl"] == pytest.approx(steps[2]["kl"], rel=1e-3)
