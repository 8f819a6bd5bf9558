"""CPU: the SFT data module and flags.  Known answers worked out by hand from the reference's per-piece layout
(/root/reference/train/src/dataset/sft_dataset.py:127-176, data_utils.py:16-38) -- the reference's dataset needs the HF processor and
a tokenizer download, neither present here, so these are restatement checks, not fixture checks."""
import json
import os

import numpy as np
import pytest

from oracle import model as om
from oracle import sft as osft
from tests.helpers import to_engine_cfg
from video_r2_amd import params
from video_r2_amd.host import frontend as fe
from video_r2_amd.host import sft_dataset as hs
from video_r2_amd.sft import split_supervised


def test_media_tag_becomes_the_vision_placeholder_and_roles_pass_through():
    turns = hs.to_turns([{"from": "human", "value": "<video>\nWhat happens?"}, {"from": "gpt", "value": "A dog runs."},
                         {"from": "human", "value": "And then?"}, {"from": "assistant", "value": "It stops."}], is_video=True)
    assert turns[0] == {"role": "user", "content": "<|vision_start|><|video_pad|><|vision_end|>What happens?"}
    assert [t["role"] for t in turns] == ["user", "gpt", "user", "assistant"]      # only "human" is renamed (data_utils.py:28)
    assert hs.media_tag_to_placeholder("look\n<image>\nhere", False) == "look<|vision_start|><|image_pad|><|vision_end|>here"


def test_rows_resolve_media_and_subtitles(tmp_path):
    (tmp_path / "subs" / "a").mkdir(parents=True)
    (tmp_path / "subs" / "a" / "v1.srt").write_text("1\n00:00:00,000 --> 00:00:01,000\nhi\n")
    rows = [{"video": "a/v1.mp4", "conversations": [{"from": "human", "value": "<video>\nq"}, {"from": "gpt", "value": "a"}]},
            {"image": ["p.png", "http://x/y.png"], "conversations": [{"from": "human", "value": "<image><image>q"}, {"from": "gpt", "value": "a"}]},
            {"conversations": [{"from": "human", "value": "q"}, {"from": "gpt", "value": "a"}]}]
    (tmp_path / "d.json").write_text(json.dumps(rows))
    a = params.parse_sft_args(["--data_path", str(tmp_path / "d.json"), "--image_folder", "/data", "--video_subtitles_folder", str(tmp_path / "subs"),
                               "--video_max_pixels", "151200", "--fps", "1.0"])
    ds = hs.make_supervised_data_module(None, None, a)["train_dataset"]
    assert len(ds) == 3
    v = ds[0]["media"][0]
    assert v["video"] == "/data/a/v1.mp4" and v["subtitles"].endswith("subs/a/v1.srt") and v["max_pixels"] == 151200 and v["fps"] == 1.0
    im = ds[1]["media"]
    assert [e["image"] for e in im] == ["/data/p.png", "http://x/y.png"] and ds[1]["is_video"] is False
    assert ds[2]["media"] == [] and ds[2]["turns"][1]["content"] == "a"


class _FakeClip:
    def __init__(self, n=8, h=56, w=56):
        self.frames = np.random.default_rng(0).integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        self.fps = 2.0

    def __len__(self):
        return len(self.frames)

    def get(self, idx):
        return self.frames[np.asarray(idx)]


def test_front_end_ids_and_labels_piece_by_piece(monkeypatch):
    cfg = to_engine_cfg(om.cfg_tiny(2))
    tok = fe.ByteTokenizer(cfg)
    monkeypatch.setenv("FPS_MAX_FRAMES", "4")
    monkeypatch.setenv("WRITE_TIMESTAMPS_ON_FRAMES", "False")
    monkeypatch.setenv("WRITE_SUBTITLES_ON_FRAMES", "False")
    front = hs.SFTFrontEnd(cfg, tok, clip_opener=lambda path: _FakeClip())
    ex = {"turns": hs.to_turns([{"from": "human", "value": "<video>\nWhat?"}, {"from": "gpt", "value": "Dog."},
                                {"from": "human", "value": "Then?"}, {"from": "gpt", "value": "Stop."}], True),
          "media": [{"type": "video", "video": "x.npz", "min_pixels": 3136, "max_pixels": 3136, "fps": 2.0, "subtitles": None}], "is_video": True}
    out = front(ex)
    frames, dims = out["videos"][0]
    n_tok = front._tokens_of(frames.shape[0], *dims)
    enc = tok.encode
    pieces = [(enc("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"), False),
              (enc("<|im_start|>user\n<|vision_start|>") + [cfg.video_token_id] * n_tok + enc("<|vision_end|>What?<|im_end|>\n<|im_start|>gpt\n"), False),
              (enc("Dog.<|im_end|>\n"), True),
              (enc("<|im_start|>user\nThen?<|im_end|>\n<|im_start|>gpt\n"), False),
              (enc("Stop.<|im_end|>\n"), True)]
    ids, labels = osft.labels_of(pieces)
    assert np.array_equal(out["input_ids"], ids) and np.array_equal(out["labels"], labels)
    assert int((out["labels"] != -100).sum()) == len(enc("Dog.<|im_end|>\n")) + len(enc("Stop.<|im_end|>\n"))
    # the cut the trainer makes: everything before the first supervised token is the prompt (all of the video sits there)
    prompt, comp, mask = split_supervised(out["input_ids"], out["labels"])
    assert len(prompt) == len(pieces[0][0]) + len(pieces[1][0]) and len(comp) == len(ids) - len(prompt)
    assert mask.sum() == (labels != -100).sum() and mask[0] and not mask[len(pieces[2][0])]
    assert cfg.video_token_id not in comp
    # a placeholder without a media element is the HF processor's error
    bad = dict(ex, media=[])
    with pytest.raises(ValueError, match="no matching media"):
        front(bad)
    with pytest.raises(ValueError, match="placeholder"):
        front(dict(ex, turns=hs.to_turns([{"from": "human", "value": "no tag"}, {"from": "gpt", "value": "x"}], True)))


def test_sft_flags_of_the_shipped_recipe_and_the_lora_checks():
    argv = ("--use_liger False --deepspeed scripts/zero3.json --lora_enable True --use_dora False --lora_rank 64 --lora_alpha 64 --lora_dropout 0.05 "
            "--num_lora_modules -1 --model_id /m --data_path d.json --image_folder /v --freeze_vision_tower True --freeze_llm True --freeze_merger True "
            "--bf16 True --fp16 False --disable_flash_attn2 False --output_dir /o --num_train_epochs 1 --per_device_train_batch_size 1 "
            "--gradient_accumulation_steps 4 --video_max_pixels 151200 --fps 1.0 --learning_rate 1e-5 --merger_lr 1e-5 --vision_lr 2e-6 --weight_decay 0.1 "
            "--warmup_ratio 0.03 --lr_scheduler_type cosine --logging_steps 1 --tf32 False --gradient_checkpointing True --report_to tensorboard "
            "--lazy_preprocess True --save_strategy no --dataloader_num_workers 4").split() + ["--lora_namespan_exclude", "['lm_head', 'embed_tokens']"]
    a = params.parse_sft_args(argv)
    assert a.lora_enable and a.lora_rank == 64 and a.lora_alpha == 64 and a.lora_dropout == 0.05 and a.freeze_llm and a.vision_lr == 2e-6
    assert a.gradient_accumulation_steps == 4 and a.warmup_ratio == 0.03 and a.weight_decay == 0.1 and a.save_strategy == "no"
    assert set(a.ignored) >= {"deepspeed", "use_liger", "gradient_checkpointing"}
    with pytest.raises(ValueError, match="freeze_llm"):                         # train_sft.py:88-89
        params.parse_sft_args(["--lora_enable", "True", "--freeze_llm", "False"])
    with pytest.raises(ValueError, match="vision_lora"):
        params.parse_sft_args(["--vision_lora", "True"])
    with pytest.raises(NotImplementedError, match="use_dora|not built"):
        params.parse_sft_args(["--lora_enable", "True", "--freeze_llm", "True", "--use_dora", "True", "--lora_namespan_exclude", "['lm_head', 'embed_tokens']"])


def test_sft_loss_is_the_shifted_token_mean():
    import torch
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(2, 5, 7, generator=g)
    labels = torch.tensor([[-100, -100, 3, 4, -100], [-100, 1, -100, -100, 6]])
    lp = torch.log_softmax(logits, -1)
    want = -(lp[0, 1, 3] + lp[0, 2, 4] + lp[1, 0, 1] + lp[1, 3, 6]) / 4            # position t predicts token t + 1
    assert abs(float(osft.sft_loss(logits, labels)) - float(want)) < 1e-6


def test_adapter_directory_loads_as_the_merged_model(tmp_path):
    """weights.load_state_dict on a directory of LoRA adapters = the reference's load_pretrained_model(model_path, model_base)
    (/root/reference/train/src/utils.py:48-76): base weights, non_lora_state_dict.bin over them, W + alpha / r * B A on the adapted
    layers; oracle.sft.LoraOracle.merged is the same sum"""
    import torch
    from safetensors.torch import save_file
    from video_r2_amd import weights as W
    ocfg = om.cfg_tiny(2)
    w = om.make_weights(ocfg, seed=5)
    base, ad = tmp_path / "base", tmp_path / "adapters"
    base.mkdir(); ad.mkdir()
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in w.items()}, str(base / "model.safetensors"))
    lo = osft.LoraOracle(w, ocfg.text.num_hidden_layers, 8, 16, seed=1, b_std=0.02)
    sd = {}
    for k in lo.A:
        sd[f"base_model.model.{k}.lora_A.weight"] = lo.A[k].detach().contiguous()
        sd[f"base_model.model.{k}.lora_B.weight"] = lo.B[k].detach().contiguous()
    save_file(sd, str(ad / "adapter_model.safetensors"))
    (ad / "adapter_config.json").write_text(json.dumps({"peft_type": "LORA", "r": 8, "lora_alpha": 16, "bias": "none", "base_model_name_or_path": str(base)}))
    new_ln = torch.full_like(w["visual.merger.ln_q.weight"], 0.5)
    torch.save({"base_model.model.visual.merger.ln_q.weight": new_ln}, str(ad / "non_lora_state_dict.bin"))
    got = W.load_state_dict(str(ad))
    want = lo.merged(w)
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.up_proj.weight", "model.layers.1.mlp.down_proj.weight"):
        assert torch.equal(got[k], want[k].to(torch.bfloat16)) and not torch.equal(got[k], w[k].to(torch.bfloat16))
    assert torch.equal(got["visual.merger.ln_q.weight"].float(), new_ln) and torch.equal(got["lm_head.weight"].float(), w["lm_head.weight"])
    assert set(got) == set(w)
    (ad / "adapter_config.json").write_text(json.dumps({"peft_type": "LORA", "r": 8, "lora_alpha": 16, "base_model_name_or_path": "/nonexistent"}))
    with pytest.raises(FileNotFoundError, match="base model"):
        W.load_state_dict(str(ad))
    assert set(W.load_state_dict(str(ad), model_base=str(base))) == set(w)


def test_sft_row_order_shards_batches_over_ranks_and_resumes():
    """iter_steps is host logic: per epoch a permutation of (seed + epoch), consecutive batches of per_device_train_batch_size rows,
    batch k to rank k mod world, `accum` of a rank's batches per optimiser step; a resumed trainer skips the steps it has done"""
    from types import SimpleNamespace
    from video_r2_amd.sft import QwenSFTTrainer

    def make(rank, world, done=0, n=22, bs=2, accum=2, steps=5):
        t = QwenSFTTrainer.__new__(QwenSFTTrainer)
        t.dataset, t.args = list(range(n)), SimpleNamespace(seed=42)
        t.rank, t.world, t.bs, t.accum, t.total_steps, t.global_step = rank, world, bs, accum, steps, done
        return t
    r0, r1 = list(make(0, 2).iter_steps()), list(make(1, 2).iter_steps())
    assert len(r0) == len(r1) == 5 and all(len(s) == 2 and all(len(mb) == 2 for mb in s) for s in r0 + r1)
    # 22 rows / (2 x 2 x 2) = 2 whole steps per epoch (the 6 leftover rows of an epoch are dropped): steps 0-1 epoch 0, 2-3 epoch 1, 4 epoch 2
    for e, (a, b) in enumerate([(0, 2), (2, 4)]):
        seen = [i for s in r0[a:b] + r1[a:b] for mb in s for i in mb]
        import torch
        order = torch.randperm(22, generator=torch.Generator().manual_seed(42 + e)).tolist()
        assert sorted(seen) == sorted(order[:16]) and len(set(seen)) == 16
        assert r0[a][0] == order[0:2] and r1[a][0] == order[2:4] and r0[a][1] == order[4:6]        # batch k -> rank k mod 2
    assert list(make(0, 2, done=3).iter_steps()) == r0[3:]
    assert list(make(0, 1, n=3, bs=2, accum=2, steps=4).iter_steps()) == []                          # fewer rows than one step: nothing to do


def test_oracle_sft_loss_and_gradients_reproduce_the_hf_fixture(golden_dir):
    """tests/golden/sft_tiny_g2.npz (oracle/make_golden.py case_sft): HF transformers' own loss for `labels` and HF autograd's gradients
    on a video row with two supervised answer spans -- the oracle's forward + oracle.sft.sft_loss + torch autograd must reproduce them.
    This pins the SFT half of the oracle against the library that holds the reference's arithmetic."""
    import torch
    from tests.helpers import pixels_from_frames
    d = np.load(os.path.join(golden_dir, "sft_tiny_g2.npz"))
    ocfg = om.cfg_tiny(int(d["group"]))
    w = om.make_weights(ocfg, seed=int(d["seed"]))
    px, grid = pixels_from_frames(d["frames_u8"], ocfg.vision)
    assert tuple(grid) == tuple(d["video_grid_thw"][0])
    wt = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ids, labels = torch.from_numpy(d["input_ids"])[None], torch.from_numpy(d["labels"])[None]
    loss = osft.sft_loss(om.forward_logits(wt, ocfg, ids, None, torch.from_numpy(px), d["video_grid_thw"], [float(d["sec"])]), labels)
    loss.backward()
    assert abs(float(loss.detach()) - float(d["hf_loss"])) < 1e-5
    for key in d.files:
        if key.startswith("hf_grad:"):
            name, ref = key[len("hf_grad:"):], torch.from_numpy(d[key])
            got = wt[name].grad[:16] if ref.dim() == 2 else wt[name].grad
            assert (got - ref).norm() <= 1e-4 * ref.norm(), name
    ref = torch.from_numpy(d["hf_lm_head_grad_rows"])
    assert (wt["lm_head.weight"].grad[:64] - ref).norm() <= 1e-4 * ref.norm()
