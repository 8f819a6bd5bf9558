"""CPU: host-side product logic (flags, reward plugins, advantages) against the oracle / hand-computed answers."""
import numpy as np
import pytest
import torch

from oracle import grpo as og
from video_r2_amd import params, rewards
from video_r2_amd.host import grpo as hg


def test_train_grpo_sh_flags_parse():
    argv = ("--deepspeed scripts/zero3.json --model_id /m --data_path d.json --freeze_vision_tower True --freeze_llm False "
            "--freeze_merger True --bf16 True --num_generations 8 --beta 0.04 --max_completion_length 1024 "
            "--max_prompt_length 32768 --reward_func_names accuracy_reward,format_reward --reward_weights 1.0,0.5 "
            "--video_max_pixels 151200 --learning_rate 1e-6 --gradient_checkpointing True --brand_new_flag 7").split()
    a = params.parse_args(argv)
    assert a.num_generations == 8 and a.beta == 0.04 and a.max_completion_length == 1024 and a.freeze_vision_tower is True
    assert a.freeze_llm is False and a.learning_rate == 1e-6 and a.video_max_pixels == 151200
    assert a.ignored["deepspeed"] == "scripts/zero3.json" and a.ignored["brand_new_flag"] == "7"
    with pytest.raises(ValueError):
        params.parse_args(["--num_generations", "1"])


def test_trainable_set_follows_the_freeze_flags():
    """reference train_grpo.py:48-64: model.visual trains unless --freeze_vision_tower, its merger unless --freeze_merger; the one
    combination this engine cannot run (vision side trainable, LM frozen) is refused, not downgraded"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vr2_train_entry", os.path.join(root, "train", "src", "train", "train_grpo.py"))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    base = "--model_id /m --data_path d.json".split()
    for flags, want in (("--freeze_vision_tower True --freeze_merger True", (False, False)), ("--freeze_vision_tower False --freeze_merger True", (True, False)),
                        ("--freeze_vision_tower True --freeze_merger False", (False, True)), ("", (True, True))):
        a = params.parse_args(base + flags.split())
        assert entry.check_trainable_set(a) == want and (a.train_vision_tower, a.train_merger) == want
    # --freeze_llm True under a trainable merger: accepted (the LM's backward then computes input gradients only)
    assert entry.check_trainable_set(params.parse_args(base + "--freeze_llm True --freeze_merger False --freeze_vision_tower True".split())) == (False, True)
    entry.check_trainable_set(params.parse_args(base + "--freeze_llm True --freeze_merger True --freeze_vision_tower True".split()))


def test_train_grpo_sh_emits_flags_the_parser_accepts(tmp_path):
    """run train/scripts/train_grpo.sh with `python` shimmed by a script that prints its argv: the launcher line must name
    the entry point, and every flag the recipe passes must parse (engine-irrelevant ones land in `ignored`)"""
    import os
    import shlex
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = tmp_path / "python"
    shim.write_text("#!/bin/bash\nprintf '%q ' \"$@\"\n")
    shim.chmod(0o755)
    env = dict(os.environ, PATH=f"{tmp_path}:{os.environ['PATH']}", NUM_DEVICES="2", MODEL_NAME="/m", DATA_PATH="d.json")
    out = subprocess.run(["bash", os.path.join(root, "train", "scripts", "train_grpo.sh"), "--max_steps", "3"], env=env,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    argv = shlex.split(out.stdout)
    assert argv[:2] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in argv and "127.0.0.1" in argv
    i = argv.index("src/train/train_grpo.py")
    a = params.parse_args(argv[i + 1:])
    assert a.model_id == "/m" and a.num_generations == 8 and a.max_completion_length == 1024 and a.beta == 0.04
    assert a.video_max_pixels == 360 * 420 and a.max_steps == 3 and a.freeze_llm is False and a.freeze_vision_tower is True
    assert a.reward_func_names == "accuracy_reward,format_reward,temporal_grounding_sentence_embedding_consistency_reward"
    assert a.reward_weights == "1.0,1.0,1.0" and "deepspeed" in a.ignored and a.fps == 2.0 and a.save_steps == 500


def test_reward_plugin_discovery_and_shipped_functions():
    a = params.parse_args(["--reward_func_names", "format_reward,accuracy_reward", "--reward_weights", "0.5,2.0"])
    funcs, w, names = rewards.load_reward_funcs("video_r2_amd.rewards", a)
    assert names == ["format_reward", "accuracy_reward"] and w == [0.5, 2.0]
    with pytest.raises(ValueError):
        rewards.load_reward_funcs("video_r2_amd.rewards", params.parse_args(["--reward_func_names", "nope_reward", "--reward_weights", "1"]))
    with pytest.raises(ValueError):
        rewards.load_reward_funcs("video_r2_amd.rewards", params.parse_args(["--reward_func_names", "format_reward", "--reward_weights", "1,2"]))
    comp = lambda s: [{"role": "assistant", "content": s}]
    cs = [comp("<think>a</think> <answer>B</answer>"), comp("<answer>B</answer>"), comp("<think>x</think><answer>3.14</answer>tail")]
    assert rewards.format_reward(completions=cs) == [1.0, 0.0, 0.0]
    # the dataset's `assistant` column is the answer TURN ({"role", "content"}: grpo_dataset.py:157,181; reward_funcs.py:18)
    turn = lambda s: {"role": "gpt", "content": s}
    got = rewards.accuracy_reward(completions=cs, assistant=[turn("<answer>B.</answer>"), turn("<answer>C</answer>"), turn("<answer>3.14</answer>")])
    assert got == [1.0, 0.0, 1.0]
    assert rewards.accuracy_reward(completions=cs, assistant=["<answer>B</answer>", "<answer>B</answer>", "<answer>3.1</answer>"]) == [1.0, 1.0, 0.0]


def test_accuracy_reward_follows_the_reference_dispatch():
    """reward_funcs.py:17-76 / reward_utils.py:6-68 case by case: no <answer> tag extracts '' (never the whole text), the kind of
    the ground truth picks the comparison, decimals must be present on both sides or neither, > 3 decimals is a regression
    target scored 1 - relative error, <= 5 plain words is scored 1 - WER, any exception scores 0 (and is reported)."""
    comp = lambda s: [{"role": "assistant", "content": s}]
    turn = lambda s: {"role": "gpt", "content": s}
    acc = lambda pred, truth: rewards.accuracy_reward(completions=[comp(pred)], assistant=[turn(truth)])[0]
    assert rewards.extract_answer("B is right") == "" and rewards.extract_answer("<answer>\n B \n</answer>") == "B"
    assert acc("B is right", "<answer>B</answer>") == 0.0                       # untagged prediction: '' -> no letter
    assert acc("<answer>(b) because</answer>", "<answer>B</answer>") == 0.0     # letter must lead the answer: '(' does not match
    assert acc("<answer>b) because</answer>", "<answer>B. the cat</answer>") == 1.0
    assert rewards.answer_kind("1,234") == "numerical" and rewards.answer_kind("0.12345") == "regression"
    assert rewards.answer_kind("red car 12") == "OCR" and rewards.answer_kind("the long free form answer of many words, really") == "free-form"
    assert rewards.answer_kind("a long answer") == "multiple choice"        # the reference's rule: ANY answer that starts with a-e is a letter choice
    assert acc("<answer>12</answer>", "<answer>12.0</answer>") == 0.0           # decimal presence must match
    assert acc("<answer>12.004</answer>", "<answer>12.0</answer>") == 1.0       # equal at 2 decimals
    assert acc("<answer>1,234</answer>", "<answer>1,234</answer>") == 1.0
    assert acc("<answer>about 3</answer>", "<answer>3</answer>") == 0.0         # whole answer must parse as a number
    assert abs(acc("<answer>0.5</answer>", "<answer>0.40000</answer>") - 0.75) < 1e-6      # regression: 1 - |0.5-0.4|/0.4
    assert acc("<answer>red car 12</answer>", "<answer>red car 12</answer>") == 1.0
    assert abs(acc("<answer>red bus 12</answer>", "<answer>red car 12</answer>") - (1 - 1 / 3)) < 1e-9
    assert rewards.word_error_rate("a b c", "a c") == pytest.approx(1 / 3)
    assert acc("<answer>x</answer>", 5) == 0.0                                   # broken column -> exception -> 0.0


def test_reward_module_seam_and_sentence_model_passthrough(tmp_path, monkeypatch):
    """--reward_module points at the USER's module (default: src.train.reward_funcs when importable); the functions of
    the reference's TAR family are called with sentence_model= (grpo_trainer.py:878-886)"""
    import sys
    import types
    mod = types.ModuleType("user_rewards_for_test")
    seen = {}

    def temporal_grounding_sentence_embedding_consistency_reward(completions, assistant, temporal_grounding, sentence_model,
                                                                 buffer_seconds, similarity_threshold, **kwargs):
        seen.update(sentence_model=sentence_model, buffer_seconds=buffer_seconds, similarity_threshold=similarity_threshold,
                    question=kwargs["prompts"][0][1]["content"][1]["text"], tg=temporal_grounding[0])
        return [0.25] * len(completions)

    def plain_reward(completions, **kwargs):
        assert "sentence_model" not in kwargs
        return [None] * len(completions)
    mod.temporal_grounding_sentence_embedding_consistency_reward = temporal_grounding_sentence_embedding_consistency_reward
    mod.plain_reward = plain_reward
    monkeypatch.setitem(sys.modules, "user_rewards_for_test", mod)
    a = params.parse_args(["--reward_module", "user_rewards_for_test", "--reward_func_names",
                           "temporal_grounding_sentence_embedding_consistency_reward,plain_reward", "--reward_weights", "1.0,1.0",
                           "--buffer_seconds", "3", "--similarity_threshold", "0.5"])
    funcs, w, names = rewards.load_reward_funcs(a.reward_module, a)
    assert names == ["temporal_grounding_sentence_embedding_consistency_reward", "plain_reward"]
    from video_r2_amd.trainer import call_reward_funcs
    ex = {"prompt": [{"role": "system", "content": "s"}, {"role": "user", "content": [{"type": "video", "video": "v"}, {"type": "text", "text": "Q?"}]}],
          "assistant": {"role": "gpt", "content": "<answer>A</answer>"}, "temporal_grounding": {"00:01-00:02": "x"}, "temporal_captions": {}}
    out = call_reward_funcs(funcs, names, [ex], [["c0", "c1"]], a, sentence_model="SENTINEL")
    assert out.shape == (2, 2) and out[:, 0].tolist() == [0.25, 0.25] and torch.isnan(out[:, 1]).all()
    assert seen == {"sentence_model": "SENTINEL", "buffer_seconds": 3, "similarity_threshold": 0.5, "question": "Q?", "tg": {"00:01-00:02": "x"}}
    # no sentence-transformers offline: selecting a TAR reward without a model must say so, not pass None silently
    with pytest.raises(RuntimeError, match="sentence-transformers"):
        rewards.load_sentence_model(names)
    assert rewards.load_sentence_model(["format_reward"]) is None
    assert rewards.resolve_reward_module(None) == "video_r2_amd.rewards" and rewards.resolve_reward_module("x.y") == "x.y"


def test_reward_rows_are_scored_concurrently_and_in_order():
    """--reward_workers: a slow (judge-like) row-wise reward over 16 rows on 4 threads takes about a quarter of the serial time and
    returns the rows in order; dataset columns are sliced with the rows, scalars are passed through"""
    import time
    from concurrent.futures import ThreadPoolExecutor
    from video_r2_amd.trainer import call_reward_funcs

    def slow_reward(completions, assistant, buffer_seconds, **kwargs):
        assert len(assistant) == len(completions) == len(kwargs["prompts"]) and buffer_seconds == 2
        out = []
        for c, a in zip(completions, assistant):
            time.sleep(0.05)
            out.append(float(c[0]["content"]) + a)
        return out
    a = params.parse_args([])
    exs = [{"prompt": [f"p{i}"], "assistant": 100.0 * i} for i in range(4)]
    texts = [[str(4 * i + g) for g in range(4)] for i in range(4)]
    t0 = time.time()
    serial = call_reward_funcs([slow_reward], ["slow_reward"], exs, texts, a)
    t_serial = time.time() - t0
    with ThreadPoolExecutor(max_workers=4) as pool:
        t0 = time.time()
        par = call_reward_funcs([slow_reward], ["slow_reward"], exs, texts, a, pool=pool, workers=4)
        t_par = time.time() - t0
    assert torch.equal(serial, par) and serial[:, 0].tolist() == [100.0 * (r // 4) + r for r in range(16)]
    assert t_par < 0.5 * t_serial, (t_par, t_serial)


def test_judge_client_speaks_the_openai_chat_contract(monkeypatch):
    """rewards.chat() against a local stand-in for the vLLM judge (train/serve_llm: OpenAI /v1/chat/completions): request
    shape (model from SERVED_MODEL, system + user message, stream false, bearer key), response parsing, and the 10 s timeout
    turning into an exception (which reward functions map to 0.0)"""
    import http.server
    import json
    import threading
    seen = {}

    class H(http.server.BaseHTTPRequestHandler):
        def do_POST(self):
            body = json.loads(self.rfile.read(int(self.headers["Content-Length"])))
            seen.update(path=self.path, auth=self.headers.get("Authorization"), body=body)
            if body["messages"][1]["content"] == "hang":
                import time
                time.sleep(1.0)
            out = json.dumps({"choices": [{"message": {"role": "assistant", "content": " TRUE\nbecause "}}]}).encode()
            self.send_response(200); self.send_header("Content-Type", "application/json"); self.send_header("Content-Length", str(len(out)))
            self.end_headers(); self.wfile.write(out)

        def log_message(self, *a):
            pass
    srv = http.server.ThreadingHTTPServer(("127.0.0.1", 0), H)
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    try:
        monkeypatch.setenv("OPENAI_API_BASE", f"http://127.0.0.1:{srv.server_address[1]}/v1")
        monkeypatch.setenv("SERVED_MODEL", "judge-model")
        monkeypatch.delenv("OPENAI_API_KEY", raising=False)
        assert rewards.chat("you are a judge", "is it consistent?") == "TRUE\nbecause"
        assert seen["path"] == "/v1/chat/completions" and seen["auth"] == "Bearer EMPTY"
        assert seen["body"] == {"model": "judge-model", "stream": False,
                                "messages": [{"role": "system", "content": "you are a judge"}, {"role": "user", "content": "is it consistent?"}]}
        with pytest.raises(Exception):
            rewards.chat("s", "hang", timeout=0.2)
    finally:
        srv.shutdown()


@pytest.mark.parametrize("loss_type", ["bnpo", "mapo", "mapo_bnpo", "grpo"])
def test_group_advantages_match_oracle(loss_type):
    rng = np.random.default_rng(3)
    r = rng.random((12, 3)).astype(np.float32)
    r[1, 2] = np.nan; r[5, 0] = np.nan
    r[:, 0] = (r[:, 0] > 0.5)
    names = ["accuracy_reward", "format_reward", "tar_reward"]
    for scale in (True, False):
        for pr in (None, "tar_reward,accuracy_reward"):
            want = og.advantages(r, [1.0, 0.5, 2.0], 4, loss_type=loss_type, scale_rewards=scale, reward_func_names=names,
                                 positive_reinforcement=pr)
            got = hg.group_advantages(torch.from_numpy(r), [1.0, 0.5, 2.0], 4, loss_type=loss_type, scale_rewards=scale,
                                      reward_func_names=names, positive_reinforcement=pr)
            np.testing.assert_allclose(got["advantages"].numpy(), want["advantages"], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(got["rewards"].numpy(), want["rewards"], rtol=1e-6)


def test_advantages_hand_computed():
    # group of 4 rewards [1, 0, 1, 0]: mean .5, unbiased std = sqrt(1/3); A = +-0.5 / (0.57735 + 1e-4)
    a = og.advantages(np.asarray([[1.0], [0.0], [1.0], [0.0]], dtype=np.float32), [1.0], 4)["advantages"]
    np.testing.assert_allclose(a, np.asarray([1, -1, 1, -1]) * 0.5 / (np.sqrt(1 / 3) + 1e-4), rtol=1e-6)
    # MAPO with p = 1/2 -> lambda = 0 -> pure z-score with BIASED sigma = 0.5
    r = np.asarray([[1.0], [0.0], [1.0], [0.0]], dtype=np.float32)
    m = og.advantages(r, [1.0], 4, loss_type="mapo", reward_func_names=["accuracy_reward"])["advantages"]
    np.testing.assert_allclose(m, np.asarray([1, -1, 1, -1]) * 0.5 / (0.5 + np.finfo(np.float32).eps), rtol=1e-6)
    # p = 1/4 -> lambda = 1 - 4*(1/4)*(3/4) = 1/4
    r = np.asarray([[1.0], [0.0], [0.0], [0.0]], dtype=np.float32)
    m = og.advantages(r, [1.0], 4, loss_type="mapo", reward_func_names=["accuracy_reward"])["advantages"]
    mu, sig = 0.25, np.sqrt(0.1875)
    want = 0.75 * (r[:, 0] - mu) / sig + 0.25 * (r[:, 0] - mu) / mu
    np.testing.assert_allclose(m, want, rtol=1e-5)


def test_oracle_loss_gradient_matches_autograd():
    rng = np.random.default_rng(5)
    B, C = 4, 9
    lp = (-1 + rng.standard_normal((B, C))).astype(np.float32)
    ref = (lp + 0.2 * rng.standard_normal((B, C))).astype(np.float32)
    old = (lp + 0.3 * rng.standard_normal((B, C))).astype(np.float32)
    mask = (np.arange(C)[None] < np.asarray([9, 4, 1, 7])[:, None]).astype(np.int32)
    adv = rng.standard_normal(B).astype(np.float32)
    for lt in ("grpo", "bnpo", "dr_grpo", "gmpo"):
        for use_old in (False, True):
            t = torch.tensor(lp, requires_grad=True)
            o = t.detach() if not use_old else torch.tensor(old)
            kl = torch.exp(torch.tensor(ref) - t) - (torch.tensor(ref) - t) - 1
            m = torch.tensor(mask).float()
            if lt == "gmpo":
                c1 = torch.exp(((t - o) * m).sum(-1) / m.sum(-1).clamp(min=1)).unsqueeze(1).expand_as(t)
            else:
                c1 = torch.exp(t - o)
            c2 = torch.clamp(c1, 0.8, 1.28)
            A = torch.tensor(adv).unsqueeze(1)
            ptl = -torch.min(c1 * A, c2 * A) + 0.04 * kl
            if lt in ("grpo", "gmpo"):
                L = ((ptl * m).sum(-1) / m.sum(-1).clamp(min=1)).mean()
            elif lt == "bnpo":
                L = (ptl * m).sum() / m.sum().clamp(min=1)
            else:
                L = (ptl * m).sum() / (B * 16)
            L.backward()
            got = og.loss(lp, mask, adv, ref_per_token_logps=ref, old_per_token_logps=old if use_old else None, beta=0.04,
                          epsilon_low=0.2, epsilon_high=0.28, loss_type=lt, max_completion_length=16)
            assert abs(got["loss"] - L.item()) < 1e-6
            np.testing.assert_allclose(got["dloss_dlogp"], t.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_loss_normalisation_unit_emulates_eight_ranks_of_one_completion():
    """The reference recipe (train_grpo.sh:4-7: 8 ranks x per_device_train_batch_size 1) computes the loss of ONE completion per
    rank (grpo_trainer.py:1146-1151) and DDP averages the ranks' gradients.  Emulated here with torch autograd: 8 independent
    single-row losses, mean of them; norm_rows=1 must reproduce value and gradient for every aggregation, `bnpo` then equals
    the `grpo` aggregation (per-sequence token mean), and the whole-batch token mean (norm_rows=0) differs on ragged lengths."""
    rng = np.random.default_rng(8)
    B, C = 8, 12
    lp = (-1 + rng.standard_normal((B, C))).astype(np.float32)
    ref = (lp + 0.2 * rng.standard_normal((B, C))).astype(np.float32)
    lens = np.asarray([12, 3, 1, 7, 12, 5, 9, 2])
    mask = (np.arange(C)[None] < lens[:, None]).astype(np.int32)
    adv = rng.standard_normal(B).astype(np.float32)
    for lt in ("grpo", "bnpo", "dr_grpo"):
        t = torch.tensor(lp, requires_grad=True)
        total = 0.0
        for rnk in range(B):                         # one rank = one completion
            row, m = t[rnk:rnk + 1], torch.tensor(mask[rnk:rnk + 1]).float()
            rr = torch.tensor(ref[rnk:rnk + 1])
            kl = torch.exp(rr - row) - (rr - row) - 1
            c1 = torch.exp(row - row.detach())
            A = torch.tensor(adv[rnk:rnk + 1]).unsqueeze(1)
            ptl = -torch.min(c1 * A, torch.clamp(c1, 0.8, 1.2) * A) + 0.04 * kl
            if lt == "grpo":
                L = ((ptl * m).sum(-1) / m.sum(-1).clamp(min=1)).mean()
            elif lt == "bnpo":
                L = (ptl * m).sum() / m.sum().clamp(min=1)
            else:
                L = (ptl * m).sum() / (1 * 16)
            total = total + L / B                    # DDP: mean over ranks
        total.backward()
        got = og.loss(lp, mask, adv, ref_per_token_logps=ref, beta=0.04, loss_type=lt, max_completion_length=16, norm_rows=1)
        assert abs(got["loss"] - total.item()) < 1e-6
        np.testing.assert_allclose(got["dloss_dlogp"], t.grad.numpy(), rtol=1e-4, atol=1e-7)
    per_rank = og.loss(lp, mask, adv, ref_per_token_logps=ref, beta=0.04, loss_type="bnpo", norm_rows=1)
    as_grpo = og.loss(lp, mask, adv, ref_per_token_logps=ref, beta=0.04, loss_type="grpo")
    whole = og.loss(lp, mask, adv, ref_per_token_logps=ref, beta=0.04, loss_type="bnpo")
    assert abs(per_rank["loss"] - as_grpo["loss"]) < 1e-6
    np.testing.assert_allclose(per_rank["dloss_dlogp"], as_grpo["dloss_dlogp"], rtol=1e-5, atol=1e-8)
    assert abs(whole["loss"] - per_rank["loss"]) > 1e-3          # long completions weigh more in the whole-batch token mean
    # the per-rank metrics are means of per-rank ratios (gather_for_metrics(...).nanmean(), :1159-1177)
    kl_rows = [og.loss(lp[i:i + 1], mask[i:i + 1], adv[i:i + 1], ref_per_token_logps=ref[i:i + 1], beta=0.04)["kl"] for i in range(B)]
    assert abs(per_rank["kl"] - np.mean(kl_rows)) < 1e-6


def test_tiny_preset_matches_the_fixture_config():
    """ModelConfig.tiny() (dry runs, rehearsals) is the same shape family the oracle's golden fixtures were made with"""
    from oracle import model as om
    from tests.helpers import to_engine_cfg
    from video_r2_amd.config import ModelConfig
    assert ModelConfig.tiny() == to_engine_cfg(om.cfg_tiny(2))


def test_key_work_covers_every_visible_pair_once():
    """vr2_attn_bwd_dkv work list: every (key row, query row, q head of the GQA group) the causal / shared-prefix mask allows is
    walked by exactly one item, whatever the cuts by query range and by heads; heaviest items come first."""
    from video_r2_amd.host import vision_plan as vp
    for (S, G, C, kt, qs, group, ms, lens) in ((70, 2, 40, 128, 48, 2, 3, None), (300, 3, 200, 128, 256, 7, 8, None), (130, 2, 64, 64, 64, 4, 64, None),
                                               (1, 1, 1, 128, 512, 7, 64, None),
                                               (70, 3, 40, 128, 48, 2, 3, [40, 7, 0]), (300, 3, 200, 128, 256, 7, 8, [129, 200, 1])):   # packed ragged rows
        ln = [C] * G if lens is None else lens
        T = S + sum(ln)
        allow = np.zeros((T, T), dtype=bool)                         # [query][key]
        allow[:S, :S] = np.tril(np.ones((S, S), dtype=bool))
        a = S
        for g in range(G):
            allow[a:a + ln[g], :S] = True
            allow[a:a + ln[g], a:a + ln[g]] = np.tril(np.ones((ln[g], ln[g]), dtype=bool))
            a += ln[g]
        kw = vp.key_work(S, G, C, k_tile=kt, q_split=qs, group=group, max_steps=ms, lens=lens)
        assert kw.dtype == np.int32 and kw.shape[1] == 8 and (kw[:, 1] <= 128).all() and (kw[:, 1] >= 1).all()
        seen = np.zeros((group, T, T), dtype=np.int32)
        steps = []
        for k0, nk, q1, l1, q2, l2, h0, nh in kw:
            assert 1 <= nh <= group and 0 <= h0 and h0 + nh <= group
            keys = np.arange(k0, k0 + nk)
            for h in range(h0, h0 + nh):
                if l1:
                    q = np.arange(q1, q1 + l1)
                    seen[h][np.ix_(q, keys)] += (keys[None, :] <= q[:, None])
                if l2:
                    seen[h][q2:q2 + l2, k0:k0 + nk] += 1
            steps.append(nh * (-(-l1 // 64) + -(-l2 // 64)))
        assert (seen == allow[None].astype(np.int32)).all()
        assert steps == sorted(steps, reverse=True)


def test_phase_timer_reports_host_phases_under_the_reference_metric_names():
    """video_r2_amd.profiling.PhaseTimer without a GPU: host phases are wall-clock intervals, summed per name; names every rank must
    report appear with 0.0 when nothing ran under them; collect() resets"""
    import time
    from video_r2_amd.profiling import PhaseTimer
    t = PhaseTimer("QwenGRPOTrainer", stream=None)
    with t.host("format_reward"):
        time.sleep(0.01)
    with t.host("format_reward"):
        time.sleep(0.01)
    with t.phase("generate"):          # no stream: falls back to the wall clock
        time.sleep(0.005)
    got = t.collect(expect=["backward"])
    assert set(got) == {"profiling/Time taken: QwenGRPOTrainer.format_reward", "profiling/Time taken: QwenGRPOTrainer.generate",
                        "profiling/Time taken: QwenGRPOTrainer.backward"}
    assert got["profiling/Time taken: QwenGRPOTrainer.format_reward"] >= 0.02 and got["profiling/Time taken: QwenGRPOTrainer.backward"] == 0.0
    assert t.collect() == {}
    off = PhaseTimer(enabled=False)
    with off.phase("x"), off.host("y"):
        pass
    assert off.collect() == {}


def test_reasoning_reward_gets_the_local_judge_and_tar_rewards_the_sentence_model():
    """the keyword contract of the reference's reward loop (grpo_trainer.py:866-892): a function named reasoning_reward is called with
    reward_tokenizer= / reward_model= (the --reward_llm_judge LM), the Temporal-Alignment family with sentence_model=, every other
    function with neither; the judge is only LOADED when reasoning_reward is selected"""
    from video_r2_amd import rewards
    from video_r2_amd.trainer import call_reward_funcs
    seen = {}

    def reasoning_reward(prompts, completions, reward_tokenizer, reward_model, **kw):
        seen["reasoning"] = (reward_tokenizer, reward_model, "sentence_model" in kw)
        return [1.0] * len(completions)

    def temporal_reward(prompts, completions, sentence_model, **kw):
        seen["tar"] = (sentence_model, "reward_model" in kw)
        return [None] * len(completions)

    def format_reward(prompts, completions, **kw):
        seen["plain"] = ("sentence_model" in kw, "reward_model" in kw)
        return [0.5] * len(completions)
    a = params.parse_args([])
    exs = [{"prompt": ["p"], "assistant": "x"}]
    out = call_reward_funcs([reasoning_reward, temporal_reward, format_reward], ["reasoning_reward", "temporal_reward", "format_reward"], exs, [["a", "b"]],
                            a, sentence_model="SM", reward_judge=("TOK", "LM"))
    assert seen == {"reasoning": ("TOK", "LM", False), "tar": ("SM", False), "plain": (False, False)}
    assert out[:, 0].tolist() == [1.0, 1.0] and torch.isnan(out[:, 1]).all() and out[:, 2].tolist() == [0.5, 0.5]
    assert rewards.load_reward_judge(["accuracy_reward", "format_reward"], "Qwen/Qwen3-0.6B") == (None, None)   # not selected: not loaded
    assert rewards.load_reward_judge(["reasoning_reward"], "") == (None, None)
