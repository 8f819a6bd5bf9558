"""Reward-function plugin surface of the GRPO trainer.

Contract kept from the reference (/root/reference/train/src/utils.py:116-157, grpo_trainer.py:846-892):
  * a reward function is any callable in a module whose name ends in ``_reward``, selected / weighted by
    ``--reward_func_names`` / ``--reward_weights``; the module is the USER's (``--reward_module``, default: the reference
    layout ``src.train.reward_funcs`` when it is importable next to the entry point, else this module);
  * it is called as ``fn(prompts=list[chat], completions=list[[{"role": "assistant", "content": str}]], **dataset_columns,
    length_reinforcement_buffer=, buffer_seconds=, similarity_threshold=[, sentence_model=])`` and returns
    ``list[float | None]`` (None -> NaN -> ignored by the nansum); the functions named in SENTENCE_MODEL_REWARDS also get the
    trainer's sentence-embedding model (grpo_trainer.py:878-886);
  * the judge is an OpenAI-compatible ``/v1/chat/completions`` endpoint at ``OPENAI_API_BASE`` serving ``SERVED_MODEL``,
    10 s timeout, failures -> reward 0.0 (reward_funcs.py:9-14, reward_utils.py:103-109).
The functions shipped here are the model-free ones (format; accuracy for multiple-choice / numeric / regression / short-text
answers, ROUGE free-form answers when ``rouge_score`` is installed).  The Temporal-Alignment reward family of the reference
(LLM judge + MiniLM matching) is the user's code and plugs in through the module seam.
"""
from __future__ import annotations

import importlib
import inspect
import json
import os
import re
import sys
import urllib.request

JUDGE_TIMEOUT_S = 10.0
# reward functions that take ``sentence_model=`` (grpo_trainer.py:878)
SENTENCE_MODEL_REWARDS = ("temporal_reward", "accuracy_model_reward", "temporal_grounding_sentence_embedding_reward",
                          "temporal_grounding_sentence_embedding_consistency_reward",
                          "temporal_grounding_sentence_embedding_f1_score_reward",
                          "temporal_grounding_sentence_embedding_f1_score_consistency_reward")
REFERENCE_MODULE = "src.train.reward_funcs"


def resolve_reward_module(requested=None):
    """--reward_module wins; otherwise the reference layout if the user's file is importable, else this module."""
    if requested:
        return requested
    try:
        importlib.import_module(REFERENCE_MODULE)
        return REFERENCE_MODULE
    except Exception:
        return __name__


def load_reward_funcs(module_path=None, training_args=None, *, name_pred=lambda n: n.endswith("_reward"), obj_pred=callable):
    module_path = resolve_reward_module(module_path)
    mod = importlib.import_module(module_path)
    available = {n: o for n, o in inspect.getmembers(mod, predicate=obj_pred) if name_pred(n)}
    if training_args is None:
        return list(available.values()), [1.0] * len(available), list(available)
    names = training_args.reward_func_names.split(",") if training_args.reward_func_names else []
    weights = [float(w) for w in training_args.reward_weights.split(",")] if training_args.reward_weights else [1.0] * len(names)
    if len(names) != len(weights):
        raise ValueError(f"Mismatch: {len(names)} reward_func_names but {len(weights)} reward_weights")
    funcs, wts, kept = [], [], []
    for name, w in zip(names, weights):
        if name not in available:
            hint = "" if module_path != __name__ else (" (this is the engine's built-in module: point --reward_module at your own, "
                                                       f"e.g. {REFERENCE_MODULE})")
            raise ValueError(f"Reward function '{name}' not found in {module_path}{hint}")
        if name == "reasoning_reward" and not getattr(training_args, "reward_llm_judge", None):
            continue
        funcs.append(available[name]); wts.append(w); kept.append(name)
    return funcs, wts, kept


def load_reward_judge(names, path):
    """The local judge LM of the reference's ``reasoning_reward`` seam (grpo_trainer.py:348-356: AutoTokenizer + AutoModelForCausalLM
    from --reward_llm_judge, handed to that function as reward_tokenizer= / reward_model=, :877-881) -- loaded only when a function of
    that name is selected (the reference loads it whenever the flag is set, used or not).  -> (tokenizer, model) or (None, None)."""
    if "reasoning_reward" not in names or not path:
        return None, None
    try:
        from transformers import AutoModelForCausalLM, AutoTokenizer
    except ImportError as e:
        raise RuntimeError("reasoning_reward needs transformers for its judge model (--reward_llm_judge)") from e
    tok = AutoTokenizer.from_pretrained(path)
    model = AutoModelForCausalLM.from_pretrained(path, torch_dtype="auto")
    model.eval()
    return tok, model


def load_sentence_model(names, model_name="all-MiniLM-L6-v2"):
    """The embedding model the TAR rewards score with (grpo_trainer.py:381-387), only when one of them is selected."""
    if not any(n in SENTENCE_MODEL_REWARDS for n in names):
        return None
    try:
        from sentence_transformers import SentenceTransformer
    except ImportError as e:
        raise RuntimeError(f"reward functions {[n for n in names if n in SENTENCE_MODEL_REWARDS]} need the sentence-transformers "
                           "package for their sentence_model= argument") from e
    return SentenceTransformer(model_name, device="cpu")


def chat(system_prompt, user_prompt, model=None, base_url=None, timeout=JUDGE_TIMEOUT_S):
    """One non-streaming chat completion against the OpenAI-compatible judge; returns the message text."""
    base = (base_url or os.environ.get("OPENAI_API_BASE", "http://localhost:8000/v1")).rstrip("/")
    body = json.dumps({"model": model or os.environ.get("SERVED_MODEL"), "stream": False,
                       "messages": [{"role": "system", "content": system_prompt}, {"role": "user", "content": user_prompt}]}).encode()
    req = urllib.request.Request(base + "/chat/completions", data=body,
                                 headers={"Content-Type": "application/json",
                                          "Authorization": "Bearer " + os.environ.get("OPENAI_API_KEY", "EMPTY")})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return (json.loads(r.read())["choices"][0]["message"]["content"] or "").strip()


# ---------------------------------------------------------------------------------------------- answer handling
def extract_answer(text):
    """Text between <answer> tags, '' when there is none (reward_utils.py:6-11)."""
    m = re.search(r"<answer>\s*(.*?)\s*</answer>", text, re.DOTALL)
    return m.group(1).strip() if m else ""


def extract_reasoning(text):
    m = re.search(r"<think>\s*(.*?)\s*</think>", text, re.DOTALL)
    return m.group(1).strip() if m else ""


def answer_kind(answer):
    """'multiple choice' | 'numerical' | 'regression' | 'OCR' | 'free-form' (reward_utils.py:52-68)."""
    a = answer.strip()
    if re.match(r"^[A-Ea-e][\.\)]?\s*(.*)", a):
        return "multiple choice"
    if re.fullmatch(r"[-+]?\d{1,3}(,\d{3})*(\.\d+)?", a) or re.fullmatch(r"[-+]?\d+(\.\d+)?", a):
        return "regression" if ("." in a and len(a.split(".")[-1]) > 3) else "numerical"
    if len(a.split()) <= 5 and re.fullmatch(r"[A-Za-z0-9\s\-:]+", a):
        return "OCR"
    return "free-form"


def option_letter(answer):
    m = re.match(r"([A-Ea-e])", answer.strip())
    return m.group(1).upper() if m else answer.strip()


def to_number(text):
    try:
        return float(text.replace(",", ""))
    except Exception as e:
        print(f"Error converting '{text}' to float: {e}")
        return None


def word_error_rate(reference, hypothesis):
    ref, hyp = reference.split(), hypothesis.split()
    prev = list(range(len(hyp) + 1))
    for i in range(1, len(ref) + 1):
        cur = [i] + [0] * len(hyp)
        for j in range(1, len(hyp) + 1):
            cur[j] = prev[j - 1] if ref[i - 1] == hyp[j - 1] else 1 + min(prev[j], cur[j - 1], prev[j - 1])
        prev = cur
    return prev[len(hyp)] / max(1, len(ref))


def rouge_mean_f(reference, hypothesis):
    """Mean of the ROUGE-1/2/L F-measures (stemmed) through rouge_score, as reward_utils.py:45-49."""
    from rouge_score import rouge_scorer
    sc = rouge_scorer.RougeScorer(["rouge1", "rouge2", "rougeL"], use_stemmer=True).score(reference, hypothesis)
    return (sc["rouge1"].fmeasure + sc["rouge2"].fmeasure + sc["rougeL"].fmeasure) / 3


def format_reward(completions, **kwargs):
    """1.0 iff the completion is exactly ``<think>...</think> <answer>...</answer>`` (reward_funcs.py:79-88)."""
    return [1.0 if re.fullmatch(r"<think>.*?</think>\s*<answer>.*?</answer>", c[0]["content"], re.DOTALL) else 0.0 for c in completions]


def accuracy_reward(completions, assistant, **kwargs):
    """reward_funcs.py:17-76: the dataset's ``assistant`` column holds the answer turn ({"role", "content"}); both sides go
    through the <answer> tag ('' without one), the kind of the GROUND TRUTH picks the comparison, any exception scores 0."""
    out = []
    for comp, turn in zip(completions, assistant):
        try:
            truth = extract_answer(turn["content"] if isinstance(turn, dict) else turn)
            pred = extract_answer(comp[0]["content"])
            kind = answer_kind(truth)
            if kind == "multiple choice":
                score = 1.0 if option_letter(pred) == option_letter(truth) else 0.0
            elif kind == "numerical":
                if (("." in truth) or ("," in truth)) != (("." in pred) or ("," in pred)):
                    score = 0.0
                else:
                    t, p = to_number(truth), to_number(pred)
                    score = 1.0 if (t is not None and p is not None and round(t, 2) == round(p, 2)) else 0.0
            elif kind == "OCR":
                score = max(0.0, min(1.0, 1 - word_error_rate(truth, pred)))
            elif kind == "free-form":
                score = max(0.0, min(1.0, rouge_mean_f(truth, pred)))
            else:  # regression
                t, p = to_number(truth), to_number(pred)
                score = 0.0 if (t is None or p is None) else max(0.0, min(1.0, 1 - (abs(p - t) + 1e-9) / (abs(t) + 1e-9)))
        except Exception as e:
            print(f"Error in reward_fn for inferred question_type: {e}", file=sys.stderr)
            score = 0.0
        out.append(score)
    return out
