"""GPU: the evaluation-harness adapter.  `generate_until` with stand-in requests shaped like the harness's
Instance.args (/root/reference/eval/lmms-eval/lmms_eval/models/qwen2_5_vl.py:181-205): answers come back in request order, the
greedy tokens equal a direct engine decode of the same prompt, `until` strings cut the text, unsupported asks fail loudly."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests.helpers import to_engine_cfg

pytestmark = pytest.mark.gpu


class Req:
    def __init__(self, *args):
        self.args = args


def test_generate_until_matches_direct_decode_and_keeps_order(tmp_path):
    from video_r2_amd.eval_adapter import VideoR2EvalModel, eval_frame_plan
    from video_r2_amd.host import processor as hp
    from video_r2_amd.host.frontend import VIDEO_PAD, ArrayClip, ByteTokenizer, render_chat
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    cfg = to_engine_cfg(ocfg)
    m = VideoR2Model.from_state_dict(cfg, om.make_weights(ocfg, seed=17), device="cuda:0")
    tok = ByteTokenizer(cfg)
    rng = np.random.default_rng(3)
    paths = []
    for i, (t, h, w) in enumerate(((40, 56, 84), (9, 56, 56))):
        p = tmp_path / f"v{i}.npz"
        np.savez(p, frames=rng.integers(0, 256, size=(t, h, w, 3), dtype=np.uint8), fps=np.float32(10.0))
        paths.append(str(p))
    docs = {"videomme": {"test": [{"video": paths[0], "q": "What happens first?"}, {"video": paths[1], "q": "How many?"}]}}
    ev = VideoR2EvalModel(model=m, tokenizer=tok, task_dict=docs, max_num_frames=6, min_pixels=4704, max_pixels=4704,
                          reasoning_prompt="\\nThink first.")
    vis = lambda doc: [doc["video"]]
    kw = {"max_new_tokens": 9, "temperature": 0, "until": ["\n\n"]}
    reqs = [Req(docs["videomme"]["test"][i]["q"] + "<image>", kw, vis, i, "videomme", "test") for i in (1, 0, 1)]
    out = ev.generate_until(reqs)
    assert len(out) == 3 and all(isinstance(a, str) for a in out)
    assert out[0] == out[2]                                               # same document, greedy: same answer, request order kept
    # direct path for document 0: same frames, same prompt, engine prefill + greedy decode
    ele = ev._video_element(paths[0])
    clip = ArrayClip(paths[0])
    idx, sfps, dims = eval_frame_plan(ele, len(clip), clip.fps, 56, 84, 6)
    assert len(idx) == 6 and idx[0] == 0 and idx[-1] == 39 and dims == (56, 84)   # 6 of the 8 stage-one frames, truncating linspace
    frames = torch.from_numpy(np.asarray(clip.get(idx.tolist()))).permute(0, 3, 1, 2).contiguous()
    (emb, plan), = m.engine.encode_videos([dict(frames_u8=frames, resize_to=dims)])
    text = render_chat([{"role": "system", "content": "You are a helpful assistant."},
                        {"role": "user", "content": [ele, {"type": "text", "text": "What happens first?\nThink first."}]}])
    a, b = text.split(VIDEO_PAD)
    n_tok = ev.front.video_tokens(6, *dims)
    ids = np.asarray(tok.encode(a) + [cfg.video_token_id] * n_tok + tok.encode(b), dtype=np.int64)
    st = m.engine.prefill(m.text, ids, emb.clone(), plan, [hp.second_per_grid_t(cfg.vision.temporal_patch_size, sfps)])
    toks, _ = m.engine.decode(m.text, st, 1, 9, temperature=1.0, top_k=1, seed=0, eos_token_id=cfg.eos_token_id)
    row = toks[0].cpu().numpy()
    stop = np.flatnonzero(row == cfg.eos_token_id)
    want = tok.decode((row[:stop[0]] if stop.size else row).tolist(), skip_special_tokens=True)
    assert out[1] == want
    # ... and against the ORACLE: the adapter's own answer tokens, teacher-forced through the CPU restatement of the model on the
    # same frames and prompt, are its arg-max at every step (unless the top two logits are a near tie at bf16 operand precision)
    one = ev.generate_until([Req(docs["videomme"]["test"][0]["q"] + "<image>", kw, vis, 0, "videomme", "test")])
    assert one == [want]
    ans = np.asarray(ev.last_token_rows[0], dtype=np.int64)
    assert ans.size > 0 and np.array_equal(ans, row[:ans.size])
    from tests.helpers import pixels_from_frames
    px, grid = pixels_from_frames(frames.numpy(), cfg.vision)                       # dims == the source size: no resize in between
    full = torch.from_numpy(np.concatenate([ids, ans]))[None]
    w17 = om.make_weights(ocfg, seed=17)
    with torch.no_grad():
        logits = om.forward_logits(w17, ocfg, full, np.ones_like(full.numpy()), torch.from_numpy(px), np.asarray([grid]),
                                   [hp.second_per_grid_t(cfg.vision.temporal_patch_size, sfps)])
    S0 = len(ids)
    top2 = logits[0, S0 - 1:S0 - 1 + ans.size].topk(2, -1)
    for j in range(ans.size):
        assert int(ans[j]) == int(top2.indices[j, 0]) or float(top2.values[j, 0] - top2.values[j, 1]) < 2e-2, j
    # `until` cuts at the first stop string; a string instead of a list is accepted
    if len(want) > 2:
        cut = ev.generate_until([Req("What happens first?", {**kw, "until": want[1]}, vis, 0, "videomme", "test")])
        assert cut == [want.split(want[1])[0]]
    with pytest.raises(ValueError):
        ev.generate_until([Req("q", {"num_beams": 4}, vis, 0, "videomme", "test")])
    with pytest.raises(ValueError):
        ev.generate_until([Req("q", {"until": 3}, vis, 0, "videomme", "test")])
    with pytest.raises(NotImplementedError):
        ev.generate_until([Req("q", {}, lambda d: [object()], 0, "videomme", "test")])
    with pytest.raises(NotImplementedError):
        ev.loglikelihood([])
    # batch_size 2: the requests share one vision pass, one prefill and one decode; greedy answers are the ones of batch_size 1
    ev2 = VideoR2EvalModel(model=m, tokenizer=tok, task_dict=docs, max_num_frames=6, min_pixels=4704, max_pixels=4704,
                           reasoning_prompt="\\nThink first.", batch_size=2)
    assert ev2.generate_until(reqs) == out
    # sampling path runs and honours the seed
    s1 = ev.generate_until([Req("q", {"temperature": 0.8, "max_new_tokens": 6, "seed": 5}, vis, 0, "videomme", "test")])
    s2 = ev.generate_until([Req("q", {"temperature": 0.8, "max_new_tokens": 6, "seed": 5}, vis, 0, "videomme", "test")])
    assert s1 == s2


def test_generate_until_takes_pil_images_like_the_harness(tmp_path):
    """a document whose visuals are PIL images (the harness sends them as RGB JPEG data URIs, qwen2_5_vl.py:239-245): the answer is
    the greedy decode of [system, user[image, image, text]] with the JPEG round trip's pixels under fetch_image's resize"""
    import base64
    import io
    from PIL import Image
    from video_r2_amd.eval_adapter import VideoR2EvalModel
    from video_r2_amd.host.frontend import IMAGE_PAD, ByteTokenizer, read_image, render_chat
    from video_r2_amd.model import VideoR2Model
    ocfg = om.cfg_tiny(2)
    cfg = to_engine_cfg(ocfg)
    m = VideoR2Model.from_state_dict(cfg, om.make_weights(ocfg, seed=19), device="cuda:0")
    tok = ByteTokenizer(cfg)
    rng = np.random.default_rng(5)
    imgs = [Image.fromarray(rng.integers(0, 256, size=(60, 90, 3), dtype=np.uint8)),
            Image.fromarray(rng.integers(0, 256, size=(64, 64, 4), dtype=np.uint8), mode="RGBA")]
    docs = {"mmbench": {"test": [{"imgs": imgs, "q": "Which one is brighter?"}]}}
    ev = VideoR2EvalModel(model=m, tokenizer=tok, task_dict=docs, min_pixels=3136, max_pixels=4704)
    out = ev.generate_until([Req("<image><image>Which one is brighter?", {"max_new_tokens": 7, "temperature": 0}, lambda d: d["imgs"], 0, "mmbench", "test")])
    assert len(out) == 1 and isinstance(out[0], str)
    # the same request by hand
    eles, clips, counts = [], [], []
    for im in imgs:
        buf = io.BytesIO()
        im.convert("RGB").save(buf, format="JPEG")
        ele = {"type": "image", "image": "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode("utf-8"), "max_pixels": 4704, "min_pixels": 3136}
        fr = read_image(ele)
        eles.append(ele); clips.append(dict(frames_u8=fr, resize_to=None)); counts.append(ev.front.video_tokens(1, int(fr.shape[2]), int(fr.shape[3])))
    (emb, plan), = m.engine.encode_videos(clips, per_prompt=[2])
    text = render_chat([{"role": "system", "content": "You are a helpful assistant."},
                        {"role": "user", "content": eles + [{"type": "text", "text": "Which one is brighter?"}]}])
    a, b, c = text.split(IMAGE_PAD)
    ids = np.asarray(tok.encode(a) + [cfg.image_token_id] * counts[0] + tok.encode(b) + [cfg.image_token_id] * counts[1] + tok.encode(c), dtype=np.int64)
    st = m.engine.prefill(m.text, ids, emb, plan, [None, None])
    toks, _ = m.engine.decode(m.text, st, 1, 7, temperature=1.0, top_k=1, seed=0, eos_token_id=cfg.eos_token_id)
    row = toks[0].cpu().numpy()
    stop = np.flatnonzero(row == cfg.eos_token_id)
    assert out[0] == tok.decode((row[:stop[0]] if stop.size else row).tolist(), skip_special_tokens=True)
