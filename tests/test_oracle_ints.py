"""CPU: the oracle's integer restatements and the product's host index code against the golden vectors
(tests/golden/indices.npz was produced from transformers.vision_utils + torch by oracle/make_golden.py) and against
the reference's own docstring known-answer examples."""
import os

import numpy as np
import pytest
import torch

from oracle import ints, vision_index as ovi
from video_r2_amd.host import rope_index as ri
from video_r2_amd.host import sampler as hs
from video_r2_amd.host import frames as hf
from video_r2_amd.host import vision_plan as vp


@pytest.fixture(scope="module")
def idx(golden_dir):
    return np.load(os.path.join(golden_dir, "indices.npz"))


def test_window_index_and_cu_seqlens(idx):
    i = 0
    while f"g{i}_grid" in idx:
        g = [tuple(int(x) for x in idx[f"g{i}_grid"])]
        for impl in (ovi.vision_window_index, vp.window_index):
            wi, cw = impl(g)
            assert np.array_equal(wi, idx[f"g{i}_window_index"]) and np.array_equal(cw, idx[f"g{i}_cu_window"]), (impl, g)
        assert np.array_equal(ovi.vision_cu_seqlens(g), idx[f"g{i}_cu_full"])
        assert np.array_equal(vp.full_cu_seqlens(g), idx[f"g{i}_cu_full"])
        assert np.array_equal(ovi.vision_position_ids(g), idx[f"g{i}_pos"])
        assert np.array_equal(vp.position_ids_hw(g), idx[f"g{i}_pos"])
        i += 1
    assert i >= 7
    for impl in (ovi.vision_window_index, vp.window_index):
        wi, cw = impl(idx["multi_grid"].tolist())
        assert np.array_equal(wi, idx["multi_window_index"]) and np.array_equal(cw, idx["multi_cu_window"])


def test_session_kats_from_survey():
    # SURVEY.md 8(c) known answers
    wi, cw = ovi.vision_window_index([(2, 16, 16)])
    assert wi[:20].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19, 24, 25, 26, 27, 4, 5, 6, 7]
    assert len(cw) - 1 == 8 and set(np.diff(cw)) == {64}
    assert ovi.vision_cu_seqlens([(2, 16, 16)]).tolist() == [0, 256, 512]
    wi, cw = ovi.vision_window_index([(16, 20, 36)])
    assert len(cw) - 1 == 240 and set(np.diff(cw).tolist()) <= {16, 32, 64}
    assert ovi.vision_position_ids([(1, 4, 4)])[:6].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1], [0, 2], [0, 3]]


def test_frame_indices_match_torch_linspace(idx):
    a = idx["linspace_cases"].tolist()
    p = 0
    n_cases = 0
    while p < len(a):
        tot, n = int(a[p]), int(a[p + 1])
        want = [int(x) for x in a[p + 2:p + 2 + n]]
        assert ints.frame_indices(tot, n) == want
        assert hf.frame_indices(tot, n).tolist() == want
        p += 2 + n
        n_cases += 1
    assert n_cases == 9
    assert ints.frame_indices(300, 32)[:8] == [0, 10, 19, 29, 39, 48, 58, 68]
    assert ints.frame_indices(300, 32)[-3:] == [280, 289, 299]


def test_repeat_sampler_golden_and_docstring_structure(idx):
    for key in idx.files:
        if not key.startswith("randperm_"):
            continue
        _, seed, n = key.split("_")
        seed, n = int(seed), int(n)
        perm = idx[key].tolist()
        got = ints.repeat_sampler(n, mini_repeat_count=2, batch_size=3, repeat_count=2, seed=seed)
        prod = list(hs.RepeatSampler(range(n), mini_repeat_count=2, batch_size=3, repeat_count=2, seed=seed))
        want = []
        for c in range(n // 3):
            for _ in range(2):
                for i in perm[3 * c:3 * c + 3]:
                    want += [i, i]
        assert got == want and prod == want
    # the reference docstring example (grpo_trainer.py:91-104): structure for 7 items, mini 2, batch 3, repeat 4
    out = ints.repeat_sampler(7, 2, 3, 4, seed=0)
    assert len(out) == 2 * 3 * 4 * 2            # incomplete third chunk dropped
    first = out[:6]
    assert out[:24] == first * 4 and first[0::2] == first[1::2] and len(set(out)) == 6
    assert len(hs.RepeatSampler(range(7), 2, 3, 4)) == 7 * 2 * 4   # __len__ as the reference (:166-167)


def test_split_tensor_dict_docstring_example():
    # grpo_trainer.py:204-214, exact
    x = torch.arange(12).reshape(6, 2)
    y = torch.arange(6).reshape(6, 1)
    for impl in (ints.split_tensor_dict, hs.split_tensor_dict):
        out = impl({"x": x, "y": y, "z": None}, 3)
        assert out[0]["x"].tolist() == [[0, 1], [2, 3]] and out[0]["y"].tolist() == [[0], [1]]
        assert out[2]["x"].tolist() == [[8, 9], [10, 11]] and out[2]["y"].tolist() == [[4], [5]]
        assert out[1]["z"] is None


def test_smart_resize_and_nframes():
    cases = [(224, 224, {}), (360, 640, {}), (1080, 1920, {}), (720, 1280, {"max_pixels": 360 * 420}),
             (100, 3000, {}), (28, 28, {}), (480, 854, {"max_pixels": 360 * 420, "min_pixels": 128 * 28 * 28})]
    for h, w, ele in cases:
        for n in (4, 16, 32, 64):
            a = ints.video_resize_dims(n, h, w, ele)
            b = hf.video_resize_dims(n, h, w, ele)
            assert a == b and a[0] % 28 == 0 and a[1] % 28 == 0
    # 224^2 is below the default video_min_pixels -> upscaled to 336^2
    assert ints.video_resize_dims(32, 224, 224) == (336, 336)
    # 16:9 source under train_grpo.sh budget 360*420 -> 280x504 "reference geometry"
    assert ints.video_resize_dims(32, 720, 1280, {"max_pixels": 360 * 420}) == (280, 504)
    with pytest.raises(ValueError):
        ints.smart_resize(10, 3000)
    with pytest.raises(ValueError):
        hf.smart_resize(10, 3000)
    for total, fps in [(300, 30.0), (17, 25.0), (1000, 24.0), (5, 30.0), (64, 2.0), (3, 1.0)]:
        for ele in ({}, {"fps": 2.0}, {"fps": 1.0, "max_frames": 16}, {"nframes": 8}):
            try:
                a = ints.smart_nframes(ele, total, fps)
            except ValueError:
                with pytest.raises(ValueError):
                    hf.smart_nframes(ele, total, fps)
                continue
            assert a == hf.smart_nframes(ele, total, fps) and a % 2 == 0 and 2 <= a <= total
    assert ints.smart_nframes({"fps": 2.0}, 300, 30.0) == 20
    assert ints.smart_nframes({"fps": 2.0}, 3000, 30.0) == 32   # FPS_MAX_FRAMES cap (train_grpo.sh:24)


def test_completion_mask():
    ids = np.array([[5, 9, 7, 9, 1], [1, 2, 3, 4, 5], [9, 9, 9, 9, 9], [1, 1, 1, 1, 9]])
    m, e, a = ints.completion_mask(ids, eos_token_id=9)
    assert m.tolist() == [[1, 1, 0, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0], [1, 1, 1, 1, 1]]
    assert e.tolist() == [1, 5, 0, 4] and a.tolist() == [True, False, True, True]
    m2, _, _ = ints.completion_mask(ids, 9, mask_truncated=True)
    assert m2[1].tolist() == [0, 0, 0, 0, 0]
    from video_r2_amd.host import grpo as hg
    pm, pe, pa = hg.completion_mask(torch.from_numpy(ids), 9)
    assert pm.dtype == torch.int32 and pm.tolist() == m.tolist() and pe.tolist() == e.tolist() and pa.tolist() == a.tolist()
    pm2, _, _ = hg.completion_mask(torch.from_numpy(ids), 9, mask_truncated=True)
    assert pm2.tolist() == m2.tolist()


def test_rope_index_against_hf_golden(golden_dir):
    for name, vid in [("fwd_tiny_g2", 2041), ("fwd_tiny_g2_pad", 2041), ("fwd_tiny_g7", 2041), ("fwd_tiny_g2_frac", 2041)]:
        f = np.load(os.path.join(golden_dir, name + ".npz"))
        ids, am = f["input_ids"], f["attention_mask"]
        pos, delta = ovi.rope_index(ids, am, f["video_grid_thw"], [float(f["sec"])], video_token_id=vid, image_token_id=2040)
        valid = am[0].astype(bool)
        assert np.array_equal(pos[:, 0, valid], f["hf_position_ids"][:, 0, valid])
        assert int(delta[0, 0]) == int(f["hf_rope_delta"][0, 0])
        p2, d2 = ri.rope_index_1seq(ids[0][valid], vid, 2040, f["video_grid_thw"], [float(f["sec"])])
        assert np.array_equal(p2, f["hf_position_ids"][:, 0, valid]) and d2 == int(f["hf_rope_delta"][0, 0])
    # HF get_rope_index docstring example (modeling_qwen2_5_vl.py:960-972): 3x2x2 grid, interval 50 -> temporal ids
    ids = np.array([[7] * 12])
    pos, _ = ovi.rope_index(ids, None, [(3, 2, 2)], [25.0], video_token_id=7, merge_size=1, tokens_per_second=2)
    assert pos[0, 0].tolist() == [0, 0, 0, 0, 50, 50, 50, 50, 100, 100, 100, 100]
    assert pos[1, 0].tolist() == [0, 0, 1, 1] * 3 and pos[2, 0].tolist() == [0, 1, 0, 1] * 3


def test_patchify_layout():
    rng = np.random.default_rng(0)
    fr = rng.integers(0, 256, size=(4, 3, 56, 84)).astype(np.float32)
    a, grid = ovi.patchify(fr)
    from video_r2_amd.host import processor as hp
    b, grid2 = hp.patchify(torch.from_numpy(fr))
    assert grid == grid2 == (2, 4, 6) and np.array_equal(a, b.numpy())
    # odd frame count: last frame repeated (video_processing_qwen2_vl.py:247-251)
    a3, g3 = ovi.patchify(fr[:3])
    b3, _ = hp.patchify(torch.from_numpy(fr[:3]))
    assert g3 == (2, 4, 6) and np.array_equal(a3, b3.numpy())
    na = ovi.normalize_frames(fr)
    nb = hp.normalize_frames(torch.from_numpy(fr)).numpy()
    assert np.array_equal(na, nb)


@pytest.mark.parametrize("h,w,oh,ow", [(360, 640, 280, 504), (224, 224, 336, 336), (97, 131, 56, 84), (30, 40, 28, 28), (56, 56, 56, 112),
                                          (8, 12, 28, 56), (29, 31, 28, 28), (2160, 3840, 252, 448), (1, 5, 28, 28)])
def test_resize_restatement_is_torch_bit_for_bit(h, w, oh, ow):
    """a6: the reference resizes uint8 frames with torchvision resize(BICUBIC, antialias=True), i.e. torch's uint8
    antialias kernel; the oracle restatement and the host tap tables must reproduce it exactly (down- and up-scaling)"""
    import torch.nn.functional as F
    from oracle import ints
    from video_r2_amd.host import frames as hf
    g = torch.Generator().manual_seed(h + ow)
    x = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8)
    want = F.interpolate(x, size=(oh, ow), mode="bicubic", antialias=True).numpy()
    got = ints.resize_bicubic_aa_u8(x.numpy(), oh, ow)
    assert np.array_equal(got, want)
    # host tap tables (what the device kernel consumes) == the oracle's taps
    for n_in, n_out in ((w, ow), (h, oh)):
        first, count, wt, bits = hf.resize_taps(n_in, n_out)
        taps, prec = ints._aa_int_weights(n_in, n_out)
        assert bits == prec and first.tolist() == [t[0] for t in taps] and count.tolist() == [len(t[1]) for t in taps]
        for i, (_, ws) in enumerate(taps):
            assert wt[i, :len(ws)].tolist() == ws and not wt[i, len(ws):].any()


def test_oracle_warpers_match_hf_logits_processors():
    """oracle.model.warp_logits / repetition_penalty (the checkers of the device sampler's top_p / min_p / repetition penalty) against
    the HF classes the reference's generate call reaches (transformers' TopK / TopP / MinP warpers, RepetitionPenaltyLogitsProcessor)."""
    import pytest
    import torch
    lp = pytest.importorskip("transformers.generation.logits_process")
    from oracle import model as om
    g = torch.Generator().manual_seed(11)
    x = 3.0 * torch.randn(5, 997, generator=g)
    ids = torch.randint(0, 997, (5, 40), generator=g)
    for T, k, tp, mp in [(0.9, 50, 0.9, None), (0.7, 50, 0.5, 0.05), (1.3, 20, 1.0, 0.2), (0.9, 64, 0.99, 0.01), (1.0, 7, 0.3, None)]:
        want = lp.TopKLogitsWarper(top_k=k)(ids, lp.TemperatureLogitsWarper(T)(ids, x.clone()))
        if tp < 1.0:
            want = lp.TopPLogitsWarper(top_p=tp)(ids, want)
        if mp:
            want = lp.MinPLogitsWarper(min_p=mp)(ids, want)
        got = om.warp_logits(x, T, k, tp, mp)
        assert torch.equal(torch.isinf(got), torch.isinf(want)), (T, k, tp, mp)
        keep = ~torch.isinf(want)
        assert torch.allclose(got[keep], want[keep])
    for pen in (1.3, 0.8):
        want = lp.RepetitionPenaltyLogitsProcessor(penalty=pen)(ids, x.clone())
        assert torch.equal(om.repetition_penalty(x, ids, pen), want)


@pytest.mark.parametrize("name", ["fwd_tiny_g2_img", "fwd_tiny_g2_img2"])
def test_image_prompts_oracle_and_host_layout_vs_hf_golden(golden_dir, name):
    """still images in the prompt (make_golden.py case_forward_images: HF forward with pixel_values / image_grid_thw): the oracle
    restatement reproduces HF's log-probs, the product's host-side layout (vision_layout: items in prompt order, None marks an
    image) reproduces HF's M-RoPE ids bit for bit, the image patchify equals HF's image processor output"""
    import os
    from oracle import model as om
    from video_r2_amd.host import rope_index as ri
    f = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = om.cfg_tiny(int(f["group"]))
    w = om.make_weights(cfg, seed=int(f["seed"]))
    ids, am = f["input_ids"], f["attention_mask"]
    mm = dict(pixel_values=torch.from_numpy(f["pixel_values"]), image_grid_thw=f["image_grid_thw"])
    grids, secs = [tuple(int(x) for x in g) for g in f["image_grid_thw"]], [None] * int(f["n_images"])
    if "frames_u8" in f:
        px, grid = ovi.patchify(ovi.normalize_frames(f["frames_u8"].astype(np.float32)))
        mm.update(pixel_values_videos=torch.from_numpy(px.astype(np.float32)), video_grid_thw=f["video_grid_thw"], second_per_grid_ts=[float(f["sec"])])
        grids.append(tuple(int(x) for x in f["video_grid_thw"][0])); secs.append(float(f["sec"]))
    lp = om.per_token_logps(w, cfg, torch.from_numpy(ids), am, int(f["C"]), float(f["temperature"]), **mm).numpy()
    assert np.abs(lp - f["hf_logps"]).max() < 2e-4
    keep = am[0].astype(bool)
    pos, delta, vis = ri.vision_layout(ids[0][keep], cfg.video_token_id, cfg.image_token_id, grids, secs)
    assert np.array_equal(pos, f["hf_position_ids"][:, 0][:, keep]) and delta == int(f["hf_rope_delta"][0, 0])
    at = 0
    for k in range(int(f["n_images"])):
        img = f[f"image{k}_u8"]
        px, g = ovi.patchify(ovi.normalize_frames(img.transpose(2, 0, 1)[None].astype(np.float32)))
        assert tuple(g) == grids[k] and np.abs(px - f["pixel_values"][at:at + len(px)]).max() < 1e-6      # HF normalises in two steps
        at += len(px)
