"""GPU parity of each HIP kernel family through the C ABI, against torch-CPU fp32 / the oracle on the same seeded inputs.
Tolerances are written next to each check; integer outputs must be exact."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

from video_r2_amd import lib  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    lib.load()
    return torch.device("cuda:0")


def S():
    return torch.cuda.current_stream().cuda_stream


def bf(x):
    return x.to(torch.bfloat16)


def pk(W, dev):
    """row-major [N][K] bf16 -> the engine's tile-packed HBM layout"""
    from video_r2_amd.weights import pack_tiles
    return pack_tiles(bf(W).to(dev).contiguous())


def test_gemm_layout_exact_integers(dev):
    # asymmetric small-integer operands: every product/sum is exact in fp32, so any fragment-layout mistake is a hard mismatch
    M, N, K = 200, 328, 192
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 3 == 0).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = A @ W.t() + bias
    Ad, Wd, bd = bf(A).to(dev), pk(W, dev), bias.to(dev)
    C = torch.zeros(M, N, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C, N, M, N, K, 1, S())
    assert torch.equal(C.cpu(), ref)
    C2 = torch.full((M, N), 2.0, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C2, N, M, N, K, 2, S())   # residual add
    assert torch.equal(C2.cpu(), ref + 2.0)
    C3 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, None, C3, N, M, N, K, 0, S())
    assert torch.equal(C3.float().cpu(), bf(A @ W.t()).float())


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (129, 132, 64), (1088, 4608, 3584), (512, 6840, 1280), (300, 1280, 3456)])
def test_gemm_random(dev, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(0.05 * torch.randn(N, K, generator=g))
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    C = torch.zeros(M, N, device=dev)
    lib.call("vr2_gemm_bf16", A.to(dev), K, pk(W, dev), bias.to(dev), C, N, M, N, K, 1, S())
    # fp32 accumulation of exact bf16 products: only summation-order noise, 1e-4 of the output scale
    err = (C.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err
    Cg = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_bf16", A.to(dev), K, pk(W, dev), bias.to(dev), Cg, N, M, N, K, 3, S())
    refg = torch.nn.functional.gelu(ref)
    assert (Cg.float().cpu() - refg).abs().max().item() <= 2 ** -7 * refg.abs().max().item() + 1e-3   # one bf16 rounding


@pytest.mark.parametrize("M,N,K", [(3900, 3848, 192), (4096, 3840, 64), (2050, 7000, 128)])
def test_gemm_256_tile_exact_integers(dev, M, N, K):
    """shapes the launcher routes to the 256 x 256 tile (8 waves, 4-phase counted-vmcnt pipeline): ragged M / N edges, 1-3 k-tiles
    (prologue / steady state / drain each take a different path), every epilogue; exact small-integer arithmetic"""
    tm, tn = -(-M // 256), -(-N // 256)
    assert tm * tn * M * 4 >= 3 * -(-tm * tn // 256) * 256 * tm * 256, "shape must satisfy the launcher's 256-tile rule (gemm.hip)"
    g = torch.Generator().manual_seed(K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 5 == 0).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = A @ W.t() + bias
    Ad, Wd, bd = bf(A).to(dev), pk(W, dev), bias.to(dev)
    C = torch.full((M + 1, N), -7.0, device=dev)                     # one guard row past M
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C, N, M, N, K, 1, S())
    assert torch.equal(C[:M].cpu(), ref) and (C[M] == -7.0).all()
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C, N, M, N, K, 2, S())   # residual add on top
    assert torch.equal(C[:M].cpu(), 2 * ref) and (C[M] == -7.0).all()
    C3 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, None, C3, N, M, N, K, 0, S())
    assert torch.equal(C3.float().cpu(), bf(A @ W.t()).float())


def test_gemm_256_tile_swiglu_and_random(dev):
    from video_r2_amd.weights import interleave_gate_up
    M, K, I = 2100, 320, 3584                                           # interleaved N = 7168: 9 x 28 tiles of 256
    g = torch.Generator().manual_seed(11)
    A = bf(torch.randn(M, K, generator=g))
    Wg = bf(0.1 * torch.randn(I, K, generator=g)); Wu = bf(0.1 * torch.randn(I, K, generator=g))
    bg = torch.randn(I, generator=g); bu = torch.randn(I, generator=g)
    Wi = interleave_gate_up(Wg, Wu); bi = interleave_gate_up(bg, bu)
    out = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    gu = torch.zeros(M, 2 * I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_swiglu_save", A.to(dev), K, pk(Wi, dev), bi.to(dev), out, I, gu, 2 * I, M, Wi.shape[0], K, S())
    pre_g = A.float() @ Wg.float().t() + bg
    pre_u = A.float() @ Wu.float().t() + bu
    ref = torch.nn.functional.silu(pre_g) * pre_u
    assert (out.float().cpu() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-5
    want_gu = interleave_gate_up(pre_g.t().contiguous(), pre_u.t().contiguous()).t()
    assert (gu.float().cpu() - want_gu).abs().max().item() <= 2 ** -8 * want_gu.abs().max().item() + 1e-5
    # random fp32-output case against torch, long K (many k-tiles through the 2-buffer ring)
    M, N, K = 2304, 2560, 3584
    A = bf(torch.randn(M, K, generator=g)); W = bf(0.05 * torch.randn(N, K, generator=g))
    C = torch.zeros(M, N, device=dev)
    lib.call("vr2_gemm_bf16", A.to(dev), K, pk(W, dev), None, C, N, M, N, K, 1, S())
    ref = A.float() @ W.float().t()
    assert (C.cpu() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("K", [128, 2048])
def test_gemm_main_plus_tail_split(dev, K):
    """M = 9 whole 256-row tiles + 64 tail rows with N = 7168: the full tiling (10 x 28 = 280 tiles, 1.09 rounds) is inefficient,
    the 252 whole tiles fill one round -> 256-row kernel on rows [0, 2304) + 128-row kernel on the tail (gemm.hip launch).
    K = 2048: the fp32-output tails (56 tiles) are additionally cut into 4 k-parts, each into a slab of the registered workspace,
    summed in part order (exact on integers; on random operands identical from run to run -- no atomics)."""
    from video_r2_amd.weights import interleave_gate_up
    lib.gemm_workspace(dev)
    M, N = 2368, 7168
    g = torch.Generator().manual_seed(4)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float() + (torch.arange(N)[:, None] % 7 == 0).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = A @ W.t() + bias
    Ad, Wd, bd = bf(A).to(dev), pk(W, dev), bias.to(dev)
    C = torch.full((M + 1, N), -7.0, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C, N, M, N, K, 1, S())
    assert torch.equal(C[:M].cpu(), ref) and (C[M] == -7.0).all()
    lib.call("vr2_gemm_bf16", Ad, K, Wd, bd, C, N, M, N, K, 2, S())
    assert torch.equal(C[:M].cpu(), 2 * ref)
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_bf16", Ad, K, Wd, None, Cb, N, M, N, K, 0, S())
    assert torch.equal(Cb.float().cpu(), bf(A @ W.t()).float())
    # random operands: the split tail is deterministic (two launches, identical bits) and right; without a workspace the tail
    # rows run as full-K tiles (also right)
    Ar, Wr = bf(torch.randn(M, K, generator=g)), bf(0.05 * torch.randn(N, K, generator=g))
    refr = Ar.float() @ Wr.float().t() + bias
    Ard, Wrd = Ar.to(dev), pk(Wr, dev)
    outs = []
    for _ in range(3):
        Cr = torch.empty(M, N, device=dev)
        lib.call("vr2_gemm_bf16", Ard, K, Wrd, bd, Cr, N, M, N, K, 1, S())
        outs.append(Cr)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[0].cpu() - refr).abs().max().item() < 1e-4 * refr.abs().max().item() + 1e-4
    lib.call("vr2_set_gemm_workspace", None, 0)
    Cn = torch.empty(M, N, device=dev)
    lib.call("vr2_gemm_bf16", Ard, K, Wrd, bd, Cn, N, M, N, K, 1, S())
    lib.gemm_workspace(dev)
    assert (Cn.cpu() - refr).abs().max().item() < 1e-4 * refr.abs().max().item() + 1e-4
    assert torch.equal(Cn[:M - 64], outs[0][:M - 64])               # the whole 256-row tiles do not depend on how the tail runs
    # SwiGLU + saved pre-activations: the tail launch must offset the aux buffer too
    I = N // 2
    Wg = bf(0.1 * torch.randn(I, K, generator=g)); Wu = bf(0.1 * torch.randn(I, K, generator=g))
    X = bf(torch.randn(M, K, generator=g))
    act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    gu = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_swiglu_save", X.to(dev), K, pk(interleave_gate_up(Wg, Wu), dev), None, act, I, gu, N, M, N, K, S())
    pg, pu = X.float() @ Wg.float().t(), X.float() @ Wu.float().t()
    ref = torch.nn.functional.silu(pg) * pu
    assert (act.float().cpu() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-5
    want = interleave_gate_up(pg.t().contiguous(), pu.t().contiguous()).t()
    assert (gu.float().cpu() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-5


@pytest.mark.parametrize("M,N,K,ns", [(8, 4608, 3584, 1), (8, 3584, 18944, 4), (4, 512, 512, 2), (16, 2048, 1152, 1), (1, 20, 128, 1), (3, 48, 160 + 32, 1), (7, 152064, 3584, 1), (32, 4608, 3584, 4), (17, 528, 1152, 2),
                                      (160, 4608, 3584, 4), (192, 3584, 18944, 8), (256, 3584, 3584, 4), (250, 1000, 1152, 1), (140, 528, 1152, 2), (210, 4608, 3584, 4)])
def test_gemv(dev, M, N, K, ns):
    g = torch.Generator().manual_seed(N + K)
    x = bf(torch.randn(M, K, generator=g))
    W = bf(0.05 * torch.randn(N, K, generator=g))
    ref = x.float() @ W.float().t()
    out = torch.full((ns, M, N), float("nan"), device=dev)
    lib.call("vr2_gemv_bf16", x.to(dev), K, M, pk(W, dev), N, K, out, M * N, N, ns, S())
    got = out.sum(0).cpu()
    assert (got - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


def test_gemm_swiglu_epilogue(dev):
    from video_r2_amd.weights import interleave_gate_up
    M, I, K = 150, 212, 192          # ragged I (ViT-like): padded to 224 inside the interleaved layout
    g = torch.Generator().manual_seed(8)
    A = bf(torch.randn(M, K, generator=g))
    Wg = bf(0.1 * torch.randn(I, K, generator=g)); Wu = bf(0.1 * torch.randn(I, K, generator=g))
    bg = torch.randn(I, generator=g); bu = torch.randn(I, generator=g)
    Wi = interleave_gate_up(Wg, Wu); bi = interleave_gate_up(bg, bu)
    assert Wi.shape[0] == 448
    out = torch.full((M, 256), 7.0, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_bf16", A.to(dev), K, pk(Wi, dev), bi.to(dev), out, 256, M, Wi.shape[0], K, 4, S())
    ref = torch.nn.functional.silu(A.float() @ Wg.float().t() + bg) * (A.float() @ Wu.float().t() + bu)
    got = out.float().cpu()
    assert (got[:, :I] - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-5   # one bf16 rounding
    assert (got[:, I:224] == 0).all() and (got[:, 224:] == 7.0).all()


def test_gemv_exact_and_gateup(dev):
    for Mx in (5, 27, 50, 64, 100, 128, 150, 192, 220, 256):         # every activation-block width (M <= 16, 32, 64, 128, 160, 192, 224, 256)
        _gateup_case(dev, Mx)


def _gateup_case(dev, M):
    I, K = 208, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (2 * I, K), generator=g).float()
    out = torch.zeros(1, M, 2 * I, device=dev)
    lib.call("vr2_gemv_bf16", bf(x).to(dev), K, M, pk(W, dev), 2 * I, K, out, M * 2 * I, 2 * I, 1, S())
    assert torch.equal(out[0].cpu(), x @ W.t())
    xs = bf(0.3 * torch.randn(M, K, generator=g))
    Ws = bf(0.1 * torch.randn(2 * I, K, generator=g))
    act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    from video_r2_amd.weights import interleave_gate_up
    lib.call("vr2_gemv_gateup_silu_bf16", xs.to(dev), K, M, pk(interleave_gate_up(Ws[:I], Ws[I:]), dev), I, K, act, I, S())
    gu = xs.float() @ Ws.float().t()
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert (act.float().cpu() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-4


def test_add_rmsnorm_and_casts(dev):
    M, H, ns = 9, 3584, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, H, generator=g)
    part = torch.randn(ns, M, H, generator=g)
    bias = torch.randn(H, generator=g)
    w = 1 + 0.1 * torch.randn(H, generator=g)
    xd = x.to(dev).clone()
    y = torch.zeros(M, H, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_add_rmsnorm", xd, H, part.to(dev), M * H, H, ns, bias.to(dev), w.to(dev), y, H, M, H, 1e-6, S())
    xr = x + part.sum(0) + bias
    yr = w * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert (xd.cpu() - xr).abs().max().item() < 1e-5
    assert (y.float().cpu() - yr).abs().max().item() <= 2 ** -8 * yr.abs().max().item() + 1e-5   # bf16 output rounding
    src = torch.randn(6, 1176, generator=g)
    idx = torch.tensor([5, 0, 3, 3, 1, 2], dtype=torch.int32)
    dst = torch.full((6, 1216), 7.0, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_cast_rows_f32_bf16", src.to(dev), 1176, idx.to(dev), dst, 1216, 6, 1176, 1216, S())
    assert torch.equal(dst[:, :1176].cpu(), bf(src[idx.long()])) and (dst[:, 1176:] == 0).all()


def _attn_ref(q, k, v, scale, allow):
    s = (q @ k.t()) * scale
    s = s.masked_fill(~allow, float("-inf"))
    return torch.nan_to_num(s.softmax(-1)) @ v


def test_attn_fwd128_edge_shapes(dev):
    """head_dim-128 forward (attention_fwd.hip) at the edges of its tiling: one query row, a visible-only key source without a
    causal part, 129 queries over 193 + 129 keys (partial last work item, partial last key tile in both sources), GQA group 7"""
    from video_r2_amd.host import vision_plan as vp
    hd, group, nKV = 128, 7, 1
    nH = group * nKV
    scale = hd ** -0.5
    g = torch.Generator().manual_seed(9)
    for (Sp, C) in ((0, 1), (5, 1), (193, 129), (64, 128), (300, 0)):
        nq = C if C else 7
        q = bf(torch.randn(nq, nH, hd, generator=g)); kp = bf(torch.randn(max(Sp, 1), nKV, hd, generator=g)); vp_ = bf(torch.randn(max(Sp, 1), nKV, hd, generator=g))
        k = bf(torch.randn(nq, nKV, hd, generator=g)); v = bf(torch.randn(nq, nKV, hd, generator=g))
        if C:      # prefix (all visible) + own causal rows
            work = vp.causal_work(C, 0, 0, Sp, 0, q_tile=vp.LM_Q_TILE)
            allow = torch.cat([torch.ones(C, Sp, dtype=torch.bool), torch.tril(torch.ones(C, C, dtype=torch.bool))], 1)
            keys = torch.cat([kp[:Sp], k], 0); vals = torch.cat([vp_[:Sp], v], 0)
        else:      # 7 queries that see the Sp prefix keys only
            work = np.asarray([(0, nq, 0, Sp, 0, 0, 0, 0)], dtype=np.int32)
            allow = torch.ones(nq, Sp, dtype=torch.bool)
            keys, vals = kp, vp_
        o = torch.full((nq + 1, nH * hd), 7.0, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(nq, nH, device=dev)
        lib.call("vr2_attn_tok", torch.from_numpy(np.ascontiguousarray(work)).to(dev), work.shape[0], q.reshape(nq, -1).to(dev), nH * hd,
                 kp.reshape(kp.shape[0], -1).to(dev), vp_.reshape(kp.shape[0], -1).to(dev), k.reshape(nq, -1).to(dev), v.reshape(nq, -1).to(dev),
                 nKV * hd, o, nH * hd, nH, group, hd, scale, lse, S())
        assert (o[nq] == 7.0).all()                                          # nothing written past the last query row
        oo = o[:nq].float().cpu().view(nq, nH, hd)
        for h in range(nH):
            s_ = (q[:, h].float() @ keys[:, 0].float().t()) * scale
            s_ = s_.masked_fill(~allow, float("-inf"))
            ref = s_.softmax(-1) @ vals[:, 0].float()
            assert (oo[:, h] - ref).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item()), (Sp, C, h)
            assert (lse[:, h].cpu() - torch.logsumexp(s_, -1)).abs().max().item() < 2e-3, (Sp, C, h)


@pytest.mark.parametrize("hd", [80, 128])
def test_attn_tok_varlen_and_causal(dev, hd):
    from video_r2_amd.host import vision_plan as vp
    g = torch.Generator().manual_seed(hd)
    nH, group = 4, 2
    nKV = nH // group
    scale = hd ** -0.5
    # (a) non-causal ragged segments (ViT windows: 16/32/64 and a long per-frame segment of 150)
    cu = np.array([0, 16, 48, 112, 262, 263], dtype=np.int32)
    T = int(cu[-1])
    q = bf(torch.randn(T, nH, hd, generator=g)); k = bf(torch.randn(T, nKV, hd, generator=g)); v = bf(torch.randn(T, nKV, hd, generator=g))
    work = torch.from_numpy(vp.segment_work(cu)).to(dev)
    qd, kd, vd = q.reshape(T, -1).to(dev), k.reshape(T, -1).to(dev), v.reshape(T, -1).to(dev)
    o = torch.zeros(T, nH * hd, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_attn_tok", work, work.shape[0], qd, nH * hd, kd, vd, kd, vd, nKV * hd, o, nH * hd, nH, group, hd, scale, None, S())
    o = o.float().cpu().view(T, nH, hd)
    for h in range(nH):
        for a, b in zip(cu[:-1], cu[1:]):
            ref = _attn_ref(q[a:b, h].float(), k[a:b, h // group].float(), v[a:b, h // group].float(), scale,
                            torch.ones(b - a, b - a, dtype=torch.bool))
            # P is rounded to bf16 before PV and the output to bf16: 2^-7 of the output scale
            assert (o[a:b, h] - ref).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item()), (h, a, b)
    # (b) shared prefix (all visible) + causal suffix, two sequences sharing the prefix (teacher-forced scoring shape)
    Sp, C, G = 100, 70, 2
    kp = bf(torch.randn(Sp, nKV, hd, generator=g)); vpp = bf(torch.randn(Sp, nKV, hd, generator=g))
    q = bf(torch.randn(G * C, nH, hd, generator=g)); k = bf(torch.randn(G * C, nKV, hd, generator=g)); v = bf(torch.randn(G * C, nKV, hd, generator=g))
    qt = vp.LM_Q_TILE if hd == 128 else vp.Q_TILE
    work = torch.from_numpy(np.concatenate([vp.causal_work(C, gg * C, 0, Sp, gg * C, q_tile=qt) for gg in range(G)])).to(dev)
    o = torch.zeros(G * C, nH * hd, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_attn_tok", work, work.shape[0], q.reshape(G * C, -1).to(dev), nH * hd, kp.reshape(Sp, -1).to(dev),
             vpp.reshape(Sp, -1).to(dev), k.reshape(G * C, -1).to(dev), v.reshape(G * C, -1).to(dev), nKV * hd, o, nH * hd,
             nH, group, hd, scale, None, S())
    o = o.float().cpu().view(G, C, nH, hd)
    allow = torch.cat([torch.ones(C, Sp, dtype=torch.bool), torch.tril(torch.ones(C, C, dtype=torch.bool))], 1)
    for gg in range(G):
        for h in range(nH):
            kk = torch.cat([kp[:, h // group], k[gg * C:(gg + 1) * C, h // group]]).float()
            vv = torch.cat([vpp[:, h // group], v[gg * C:(gg + 1) * C, h // group]]).float()
            ref = _attn_ref(q[gg * C:(gg + 1) * C, h].float(), kk, vv, scale, allow)
            assert (o[gg, :, h] - ref).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item())


def test_attn_online_softmax_rescale_branch(dev):
    # force the running max to jump in a LATER key tile (spiked key) so the alpha-rescale path is exercised
    from video_r2_amd.host import vision_plan as vp
    hd, T = 128, 200
    g = torch.Generator().manual_seed(9)
    q = bf(torch.randn(T, 1, hd, generator=g)); k = bf(torch.randn(T, 1, hd, generator=g)); v = bf(torch.randn(T, 1, hd, generator=g))
    k[150, 0] = bf(8.0 * q[17, 0].float())
    k[199, 0] = bf(6.0 * q[3, 0].float())
    work = torch.from_numpy(vp.segment_work(np.array([0, T], dtype=np.int32))).to(dev)
    o = torch.zeros(T, hd, dtype=torch.bfloat16, device=dev)
    qd, kd, vd = q.view(T, hd).to(dev), k.view(T, hd).to(dev), v.view(T, hd).to(dev)
    lib.call("vr2_attn_tok", work, work.shape[0], qd, hd, kd, vd, kd, vd, hd, o, hd, 1, 1, hd, hd ** -0.5, None, S())
    ref = _attn_ref(q[:, 0].float(), k[:, 0].float(), v[:, 0].float(), hd ** -0.5, torch.ones(T, T, dtype=torch.bool))
    assert (o.float().cpu() - ref).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("group,step", [(7, 0), (7, 300), (2, 63), (8, 64)])
def test_attn_decode(dev, group, step):
    hd, nKV, G, Sp, C = 128, 2, 3, 333, 512
    nH = nKV * group
    g = torch.Generator().manual_seed(step + group)
    kp = bf(torch.randn(Sp, nKV, hd, generator=g)); vp_ = bf(torch.randn(Sp, nKV, hd, generator=g))
    ks = bf(torch.randn(G * C, nKV, hd, generator=g)); vs = bf(torch.randn(G * C, nKV, hd, generator=g))
    q = bf(torch.randn(G, nH, hd, generator=g))
    chunk = 256
    npc, nsc = (Sp + chunk - 1) // chunk, (C + chunk - 1) // chunk
    nch = npc + nsc
    po = torch.zeros(G * nH * nch, hd, device=dev); pml = torch.zeros(G * nH * nch, 2, device=dev)
    out = torch.zeros(G, nH * hd, dtype=torch.bfloat16, device=dev)
    st = torch.tensor([step], dtype=torch.int32, device=dev)
    plen = torch.tensor([Sp], dtype=torch.int32, device=dev)
    lib.call("vr2_attn_decode", q.view(G, -1).to(dev), nH * hd, kp.view(Sp, -1).to(dev), vp_.view(Sp, -1).to(dev),
             ks.view(G * C, -1).to(dev), vs.view(G * C, -1).to(dev), nKV * hd, plen, G, 0, C, st, chunk, chunk, npc, nsc, G, nH, group, hd,
             hd ** -0.5, po, pml, out, nH * hd, None, S())
    out = out.float().cpu().view(G, nH, hd)
    for gg in range(G):
        for h in range(nH):
            kk = torch.cat([kp[:, h // group], ks[gg * C:gg * C + step + 1, h // group]]).float()
            vv = torch.cat([vp_[:, h // group], vs[gg * C:gg * C + step + 1, h // group]]).float()
            ref = _attn_ref(q[gg, h:h + 1].float(), kk, vv, hd ** -0.5, torch.ones(1, kk.shape[0], dtype=torch.bool))
            assert (out[gg, h] - ref[0]).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item())


def test_attn_decode_skips_the_own_keys_of_finished_sequences(dev):
    """finished_i32: a sequence that has produced its EOS no longer walks its own suffix keys (its later tokens are pad ids whatever
    the model computes): its output is the attention over the shared prefix alone, the live sequences are untouched -- also when the
    partial slots still hold an earlier step's values"""
    hd, nKV, group, G, Sp, C, step = 128, 2, 7, 3, 200, 256, 130
    nH = nKV * group
    g = torch.Generator().manual_seed(5)
    kp = bf(torch.randn(Sp, nKV, hd, generator=g)); vp_ = bf(torch.randn(Sp, nKV, hd, generator=g))
    ks = bf(torch.randn(G * C, nKV, hd, generator=g)); vs = bf(torch.randn(G * C, nKV, hd, generator=g))
    q = bf(torch.randn(G, nH, hd, generator=g))
    chunk = 64
    npc, nsc = (Sp + chunk - 1) // chunk, (C + chunk - 1) // chunk
    nch = npc + nsc
    po = torch.full((G * nH * nch, hd), 3.0, device=dev); pml = torch.full((G * nH * nch, 2), 1.0, device=dev)   # stale slots
    out = torch.zeros(G, nH * hd, dtype=torch.bfloat16, device=dev)
    st = torch.tensor([step], dtype=torch.int32, device=dev)
    plen = torch.tensor([Sp], dtype=torch.int32, device=dev)
    fin = torch.tensor([0, 1, 0], dtype=torch.int32, device=dev)
    lib.call("vr2_attn_decode", q.view(G, -1).to(dev), nH * hd, kp.view(Sp, -1).to(dev), vp_.view(Sp, -1).to(dev),
             ks.view(G * C, -1).to(dev), vs.view(G * C, -1).to(dev), nKV * hd, plen, G, 0, C, st, chunk, chunk, npc, nsc, G, nH, group, hd,
             hd ** -0.5, po, pml, out, nH * hd, fin, S())
    out = out.float().cpu().view(G, nH, hd)
    for gg in range(G):
        own = 0 if gg == 1 else step + 1
        for h in range(nH):
            kk = torch.cat([kp[:, h // group], ks[gg * C:gg * C + own, h // group]]).float()
            vv = torch.cat([vp_[:, h // group], vs[gg * C:gg * C + own, h // group]]).float()
            ref = _attn_ref(q[gg, h:h + 1].float(), kk, vv, hd ** -0.5, torch.ones(1, kk.shape[0], dtype=torch.bool))
            assert (out[gg, h] - ref[0]).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item())


def test_attn_decode_two_prompt_groups(dev):
    """two prompt groups with DIFFERENT prefix lengths batched in one launch (M = 2 groups x 3 sequences)"""
    hd, nKV, group, spg, C, step = 128, 2, 7, 3, 128, 17
    nH = nKV * group
    lens = [150, 97]
    Smax = max(lens)
    P = len(lens); G = P * spg
    g = torch.Generator().manual_seed(77)
    kp = bf(torch.randn(P, Smax, nKV, hd, generator=g)); vp_ = bf(torch.randn(P, Smax, nKV, hd, generator=g))
    ks = bf(torch.randn(G * C, nKV, hd, generator=g)); vs = bf(torch.randn(G * C, nKV, hd, generator=g))
    q = bf(torch.randn(G, nH, hd, generator=g))
    chunk = 64
    pchunk = 128
    npc, nsc = (Smax + pchunk - 1) // pchunk, (C + chunk - 1) // chunk
    nch = npc + nsc
    po = torch.zeros(G * nH * nch, hd, device=dev); pml = torch.zeros(G * nH * nch, 2, device=dev)
    out = torch.zeros(G, nH * hd, dtype=torch.bfloat16, device=dev)
    st = torch.tensor([step], dtype=torch.int32, device=dev)
    plen = torch.tensor(lens, dtype=torch.int32, device=dev)
    lib.call("vr2_attn_decode", q.view(G, -1).to(dev), nH * hd, kp.view(P * Smax, -1).to(dev), vp_.view(P * Smax, -1).to(dev),
             ks.view(G * C, -1).to(dev), vs.view(G * C, -1).to(dev), nKV * hd, plen, spg, Smax * nKV * hd, C, st, chunk, pchunk, npc, nsc, G, nH,
             group, hd, hd ** -0.5, po, pml, out, nH * hd, None, S())
    out = out.float().cpu().view(G, nH, hd)
    for gg in range(G):
        p = gg // spg
        for h in range(nH):
            kk = torch.cat([kp[p, :lens[p], h // group], ks[gg * C:gg * C + step + 1, h // group]]).float()
            vv = torch.cat([vp_[p, :lens[p], h // group], vs[gg * C:gg * C + step + 1, h // group]]).float()
            ref = _attn_ref(q[gg, h:h + 1].float(), kk, vv, hd ** -0.5, torch.ones(1, kk.shape[0], dtype=torch.bool))
            assert (out[gg, h] - ref[0]).abs().max().item() < 2 ** -6 * max(1.0, ref.abs().max().item())


def test_rope_kernels(dev):
    from oracle import model as om
    hd, nq, nkv, T = 128, 4, 2, 37
    g = torch.Generator().manual_seed(11)
    pos = torch.randint(0, 3000, (3, T), generator=g)
    cfg = om.cfg_tiny(2)
    cos, sin = om.mrope_cos_sin(cfg, pos[:, None, :])          # [1,T,128]
    inv = (1.0 / (cfg.text.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(dev)
    sec = torch.tensor([0] * 16 + [1] * 24 + [2] * 24, dtype=torch.int32, device=dev)
    cs = torch.zeros(T, 64, device=dev); sn = torch.zeros(T, 64, device=dev)
    lib.call("vr2_mrope_table", pos.to(torch.int32).to(dev), T, inv, sec, cs, sn, hd, S())
    # fp32 cos/sin of identical fp32 angles: libm-level differences only
    assert (cs.cpu() - cos[0, :, :64]).abs().max().item() < 2e-6 and (sn.cpu() - sin[0, :, :64]).abs().max().item() < 2e-6
    qkv = torch.randn(T, (nq + 2 * nkv) * hd, generator=g)
    qo = torch.zeros(T, nq * hd, dtype=torch.bfloat16, device=dev)
    kc = torch.zeros(50, nkv * hd, dtype=torch.bfloat16, device=dev); vc = torch.zeros(50, nkv * hd, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_rope_qkv", qkv.to(dev), qkv.stride(0), cs, sn, T, nq, nkv, hd, qo, nq * hd, kc, vc, nkv * hd, 5, S())
    x = qkv.view(T, nq + 2 * nkv, hd)
    rot = x[:, :nq + nkv] * cos[0][:, None] + om.rotate_half(x[:, :nq + nkv]) * sin[0][:, None]
    # fp32 rotation, ONE rounding to bf16
    assert (qo.float().cpu().view(T, nq, hd) - rot[:, :nq]).abs().max().item() <= 2 ** -8 * rot.abs().max().item()
    assert (kc.float().cpu()[5:5 + T].view(T, nkv, hd) - rot[:, nq:]).abs().max().item() <= 2 ** -8 * rot.abs().max().item()
    assert torch.equal(vc.cpu()[5:5 + T], bf(qkv[:, (nq + nkv) * hd:])) and (kc[:5] == 0).all()


@pytest.mark.parametrize("two_stage", [False, True])
def test_sample_topk_distribution_and_eos(dev, two_stage):
    V, G, k = 4096, 8, 50
    ws = torch.zeros(int(lib.load().vr2_sample_ws_bytes(G)) // 4, device=dev) if two_stage else None
    g = torch.Generator().manual_seed(2)
    logits = (3.0 * torch.randn(1, V, generator=g)).to(dev)
    T = 0.9
    kth = torch.topk(logits[0].cpu() / T, k)[0][-1]
    keep = (logits[0].cpu() / T) >= kth
    p = torch.where(keep, (logits[0].cpu() / T), torch.tensor(float("-inf"))).softmax(-1)
    nxt = torch.zeros(G, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    counts = torch.zeros(V)
    n_draw = 600
    for it in range(n_draw):
        lib.call("vr2_sample_topk", logits, 0, G, V, T, k, 1234 + it, None, step, None, -1, 0, nxt, None, 0, ws, S())
        counts += torch.bincount(nxt.cpu().long(), minlength=V).float()
    assert counts[~keep].sum().item() == 0                     # exact: nothing outside the top-k set is ever drawn
    n = counts.sum().item()
    exp = p * n
    sel = exp > 5
    chi2 = (((counts - exp) ** 2)[sel] / exp[sel]).sum().item()
    dof = int(sel.sum().item()) - 1
    assert chi2 < dof + 6 * (2 * dof) ** 0.5, (chi2, dof)      # chi-square within 6 sigma of its mean
    # reproducible per (seed, step, row); rows differ
    a = torch.zeros(G, dtype=torch.int32, device=dev); b = torch.zeros(G, dtype=torch.int32, device=dev)
    lib.call("vr2_sample_topk", logits, 0, G, V, T, k, 77, None, step, None, -1, 0, a, None, 0, ws, S())
    lib.call("vr2_sample_topk", logits, 0, G, V, T, k, 77, None, step, None, -1, 0, b, None, 0, ws, S())
    assert torch.equal(a, b) and len(set(a.tolist())) > 1
    # EOS/pad bookkeeping: greedy (top_k=1) picks argmax; rows already finished emit pad; picking EOS sets finished
    am = int(logits[0].argmax().item())
    fin = torch.tensor([0, 1, 0, 0, 0, 0, 0, 1], dtype=torch.int32, device=dev)
    toks = torch.full((G, 4), -5, dtype=torch.int32, device=dev)
    step.fill_(2)
    lib.call("vr2_sample_topk", logits, 0, G, V, T, 1, 5, None, step, fin, am, 99, a, toks, 4, ws, S())
    assert a.tolist() == [am, 99, am, am, am, am, am, 99] and fin.tolist() == [1] * 8
    assert toks[:, 2].tolist() == a.tolist() and (toks[:, [0, 1, 3]] == -5).all()


def test_sample_two_stage_equals_single_pass(dev):
    # same seed/step/row -> the two-stage kernel must pick exactly the token of the single-pass kernel (real vocab size)
    V, G = 152064, 8
    g = torch.Generator().manual_seed(12)
    logits = (2.0 * torch.randn(G, V, generator=g)).to(dev)
    ws = torch.zeros(int(lib.load().vr2_sample_ws_bytes(G)) // 4, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    a = torch.zeros(G, dtype=torch.int32, device=dev); b = torch.zeros(G, dtype=torch.int32, device=dev)
    for it in range(20):
        step.fill_(it)
        lib.call("vr2_sample_topk", logits, V, G, V, 0.9, 50, 99, None, step, None, -1, 0, a, None, 0, None, S())
        lib.call("vr2_sample_topk", logits, V, G, V, 0.9, 50, 99, None, step, None, -1, 0, b, None, 0, ws, S())
        assert torch.equal(a, b), it
    kth = torch.topk(logits.cpu(), 50)[0][:, -1]
    assert (torch.gather(logits.cpu(), 1, a.cpu().long()[:, None])[:, 0] >= kth).all()


def test_sample_two_stage_top_values_crowded_into_few_threads(dev):
    """the candidate kernel bounds a slice's k-th largest key by the k-th largest of its 256 THREAD maxima and ranks the keys above
    that bound in a short list; when the slice's large values all sit in the registers of a few threads the bound is low, the list
    overflows and the kernel must fall back to the radix select over all keys -- same token as the single-pass kernel either way"""
    V, G, NSLICE = 152064, 4, 32
    L = V // NSLICE
    g = torch.Generator().manual_seed(21)
    logits = torch.randn(G, V, generator=g)
    i = torch.arange(L)
    crowded = (i % 256) < 41                                   # thread tid holds slice elements tid + 256 j: 41 threads x 19 = 779 keys
    logits[0, :L][crowded] += 30.0                             # row 0, slice 0: 779 keys far above the rest (> the 768-entry list)
    logits[1, 5 * L:6 * L][crowded] += 30.0                    # row 1, slice 5
    logits[2, :L][(i % 256) < 3] += 30.0                       # row 2: 57 large keys in 3 threads (the list holds them: no fallback)
    logits = logits.to(dev)
    ws = torch.zeros(int(lib.load().vr2_sample_ws_bytes(G)) // 4, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    a = torch.zeros(G, dtype=torch.int32, device=dev); b = torch.zeros(G, dtype=torch.int32, device=dev)
    for it in range(16):
        step.fill_(it)
        lib.call("vr2_sample_topk", logits, V, G, V, 0.9, 50, 7, None, step, None, -1, 0, a, None, 0, None, S())
        lib.call("vr2_sample_topk", logits, V, G, V, 0.9, 50, 7, None, step, None, -1, 0, b, None, 0, ws, S())
        assert torch.equal(a, b), (it, a.tolist(), b.tolist())
    kth = torch.topk(logits.cpu(), 50)[0][:, -1]
    assert (torch.gather(logits.cpu(), 1, b.cpu().long()[:, None])[:, 0] >= kth).all()
    assert int(b[0]) < L and (int(b[0]) % 256) < 41 and 5 * L <= int(b[1]) < 6 * L


def test_lse_gather(dev):
    R, V = 33, 2048
    g = torch.Generator().manual_seed(4)
    lg = 4 * torch.randn(R, V, generator=g)
    tok = torch.randint(0, V, (R,), generator=g)
    out = torch.zeros(R, device=dev)
    lib.call("vr2_lse_gather", lg.to(dev), V, R, V, tok.to(torch.int32).to(dev), 0.9, out, None, S())
    ref = torch.gather(lg / 0.9, 1, tok[:, None])[:, 0] - torch.logsumexp(lg / 0.9, -1)
    assert rel_err(out.cpu().numpy(), ref.numpy()).max() < 1e-5


@pytest.mark.parametrize("loss_type", ["grpo", "bnpo", "dr_grpo", "gmpo", "mapo", "mapo_bnpo"])
@pytest.mark.parametrize("with_old", [False, True])
def test_grpo_loss_kernel(dev, loss_type, with_old):
    from oracle import grpo as og
    from video_r2_amd.host import grpo as hg
    B, C = 8, 37
    rng = np.random.default_rng(7)
    lp = (-2 + rng.standard_normal((B, C))).astype(np.float32)
    ref = (lp + 0.3 * rng.standard_normal((B, C))).astype(np.float32)
    old = (lp + 0.25 * rng.standard_normal((B, C))).astype(np.float32) if with_old else None
    lens = rng.integers(1, C + 1, size=B); lens[2] = C
    mask = (np.arange(C)[None, :] < lens[:, None]).astype(np.int32)
    adv = rng.standard_normal(B).astype(np.float32)
    want = og.loss(lp, mask, adv, ref_per_token_logps=ref, old_per_token_logps=old, beta=0.04, epsilon_low=0.2,
                   epsilon_high=0.28, loss_type=loss_type, max_completion_length=64)
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    got = hg.grpo_loss(t(lp), t(mask), t(adv), ref_logps=t(ref), old_logps=t(old), beta=0.04, epsilon_low=0.2,
                       epsilon_high=0.28, loss_type=loss_type, max_completion_length=64, stream=S())
    # fp32 elementwise math + different reduction order: 1e-5 relative
    assert abs(got["loss"].item() - want["loss"]) <= 1e-5 * max(1.0, abs(want["loss"]))
    assert abs(got["kl"].item() - want["kl"]) <= 1e-5 * max(1.0, abs(want["kl"]))
    for k in ("clip_low", "clip_high", "clip_region"):
        assert abs(got[k].item() - want[k]) < 1e-6
    assert np.abs(got["dlogp"].cpu().numpy() - want["dloss_dlogp"]).max() < 1e-6


@pytest.mark.parametrize("loss_type", ["bnpo", "mapo_bnpo", "grpo", "dr_grpo", "gmpo"])
@pytest.mark.parametrize("unit", [1, 2, 4])
def test_grpo_loss_kernel_per_rank_normalisation(dev, loss_type, unit):
    """norm_rows = the reference's per-rank micro-batch (equal on every rank, so it divides B): ragged lengths, an empty completion"""
    from oracle import grpo as og
    from video_r2_amd.host import grpo as hg
    B, C = 8, 41
    rng = np.random.default_rng(11 + unit)
    lp = (-2 + rng.standard_normal((B, C))).astype(np.float32)
    ref = (lp + 0.3 * rng.standard_normal((B, C))).astype(np.float32)
    old = (lp + 0.25 * rng.standard_normal((B, C))).astype(np.float32)
    lens = np.asarray([1, C, 7, 0, 23, 40, 2, 15])
    mask = (np.arange(C)[None, :] < lens[:, None]).astype(np.int32)
    adv = rng.standard_normal(B).astype(np.float32)
    want = og.loss(lp, mask, adv, ref_per_token_logps=ref, old_per_token_logps=old, beta=0.04, epsilon_low=0.2,
                   epsilon_high=0.28, loss_type=loss_type, max_completion_length=64, norm_rows=unit)
    t = lambda a: torch.from_numpy(a).to(dev)
    got = hg.grpo_loss(t(lp), t(mask), t(adv), ref_logps=t(ref), old_logps=t(old), beta=0.04, epsilon_low=0.2,
                       epsilon_high=0.28, loss_type=loss_type, max_completion_length=64, norm_rows=unit, stream=S())
    assert abs(got["loss"].item() - want["loss"]) <= 1e-5 * max(1.0, abs(want["loss"]))
    assert abs(got["kl"].item() - want["kl"]) <= 1e-5 * max(1.0, abs(want["kl"]))
    for k in ("clip_low", "clip_high", "clip_region"):
        assert abs(got[k].item() - want[k]) < 1e-6
    assert np.abs(got["dlogp"].cpu().numpy() - want["dloss_dlogp"]).max() < 1e-6


@pytest.mark.parametrize("M", [5, 16, 27, 64, 128])
def test_gemv_fp8_weights_exact_and_random(dev, M):
    """weight-only fp8 decode GEMVs (config 5): e4m3 weights with per-channel scales, bf16 activations.  Exact case: integer
    fp8 codes x power-of-two scales and small-integer activations make every product and sum exact; random case: against
    the matmul with the DEQUANTISED weights (the quantisation itself is the host's, weights.quantize_rows_fp8)"""
    from video_r2_amd.weights import dequantize_rows_fp8, interleave_gate_up, pack_tiles_fp8, quantize_rows_fp8
    g = torch.Generator().manual_seed(M)
    N, K = 424, 512                                                   # ragged N (26.5 tiles), 2 chunks of 256 k
    Wi = torch.randint(-8, 9, (N, K), generator=g).float()
    sc = 2.0 ** torch.randint(-4, 2, (N,), generator=g).float()
    q8 = Wi.to(torch.float8_e4m3fn).view(torch.uint8)
    npad = (N + 15) // 16 * 16
    spad = torch.ones(npad); spad[:N] = sc
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    for ns in (1, 2):
        out = torch.full((ns, M, N), float("nan"), device=dev)
        lib.call("vr2_gemv_w8", bf(x).to(dev), K, M, pack_tiles_fp8(q8.to(dev)), spad.to(dev), N, K, out, M * N, N, ns, S())
        assert torch.equal(out.sum(0).cpu(), x @ (Wi * sc[:, None]).t())
    # random weights through the host quantiser, K with a partial last chunk (K % 256 = 64 -> tail path), gate/up fused
    I, K2 = 208, 576
    Wg, Wu = 0.1 * torch.randn(I, K2, generator=g), 0.1 * torch.randn(I, K2, generator=g)
    Wint = interleave_gate_up(Wg, Wu)
    q, s = quantize_rows_fp8(Wint)
    Wd = dequantize_rows_fp8(q, s)
    xs = bf(0.3 * torch.randn(M, K2, generator=g))
    act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemv_gateup_silu_w8", xs.to(dev), K2, M, pack_tiles_fp8(q.to(dev)), s.to(dev), I, K2, act, I, S())
    pre = xs.float() @ Wd.t()                                          # interleaved rows: [gate tile | up tile] x 13
    pre = pre.view(M, I // 16, 2, 16)
    ref = torch.nn.functional.silu(pre[:, :, 0]) * pre[:, :, 1]
    assert (act.float().cpu().view(M, I // 16, 16) - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-4
    assert (Wd - Wint).abs().max().item() <= 2 ** -4 * Wint.abs().max().item()      # e4m3: 3 mantissa bits


def _deinterleave_fp8_rows(q8, K):
    """bytes in vr2_quant_rows_fp8 order -> natural k order: slot s of a 64-k block holds k 8s..8s+7 then 32+8s..32+8s+7"""
    M, kp = q8.shape[0], (K + 63) // 64 * 64
    return q8[:, :kp].reshape(M, kp // 64, 4, 2, 8).permute(0, 1, 3, 2, 4).reshape(M, kp)[:, :K]


@pytest.mark.parametrize("M,K", [(5, 200), (33, 1152), (128, 3584)])
def test_quant_rows_fp8_matches_torch_e4m3(dev, M, K):
    """per-row e4m3 quantisation of bf16 activations (BASELINE config 5, fp8 activations): bytes equal torch's
    float8_e4m3fn rounding of x * (448 / amax) in the kernel's fragment order, scale = amax / 448, zero rows and the zero
    padding of the last 64-k block included"""
    g = torch.Generator().manual_seed(M + K)
    x = (3 * torch.randn(M, K, generator=g)).to(torch.bfloat16)
    x[min(2, M - 1)] = 0                                                   # an all-zero row: scale 1, bytes 0
    kp = (K + 127) // 128 * 128
    q8 = torch.full((M, kp), 0xAA, dtype=torch.uint8, device=dev)
    sc = torch.zeros(M, device=dev)
    lib.call("vr2_quant_rows_fp8", x.to(dev), K, M, K, q8, kp, sc, S())
    torch.cuda.synchronize()
    xf = x.float()
    am = xf.abs().amax(1, keepdim=True)
    inv = torch.where(am > 0, torch.full_like(am, 448.0) / am, torch.ones_like(am))   # a true division (scalar / tensor is reciprocal * scalar in torch)
    want = (xf * inv).to(torch.float8_e4m3fn).view(torch.uint8)
    got = _deinterleave_fp8_rows(q8.cpu(), K)
    assert torch.equal(got, want)
    assert torch.equal(sc.cpu(), torch.where(am > 0, am / 448.0, torch.ones_like(am))[:, 0])
    k64 = (K + 63) // 64 * 64
    assert torch.all(_deinterleave_fp8_rows(q8.cpu(), k64)[:, K:] == 0)


@pytest.mark.parametrize("M", [20, 64, 128, 150, 176, 220, 256])
def test_gemv_fp8_activations_on_the_fp8_mfma(dev, M):
    """vr2_gemv_a8w8 / vr2_gemv_gateup_silu_a8w8 (v_mfma_f32_16x16x32_fp8_fp8): against the matmul of the DEQUANTISED operands in
    float64 -- the products of two e4m3 values are exact in fp32, only the summation order differs -- for a ragged N, split-K
    and the fused SiLU*mul epilogue; plus an exact-integer case"""
    from video_r2_amd.weights import dequantize_rows_fp8, interleave_gate_up, pack_tiles_fp8, quantize_rows_fp8
    g = torch.Generator().manual_seed(100 + M)
    N, K = 424, 1152
    W = torch.randn(N, K, generator=g) * 0.05
    x = (2 * torch.randn(M, K, generator=g)).to(torch.bfloat16)
    qw, sw = quantize_rows_fp8(W)
    Wd = dequantize_rows_fp8(qw, sw).double()
    q8 = torch.zeros((M, K), dtype=torch.uint8, device=dev)
    xs = torch.zeros(M, device=dev)
    lib.call("vr2_quant_rows_fp8", x.to(dev), K, M, K, q8, K, xs, S())
    xd = (_deinterleave_fp8_rows(q8.cpu(), K).view(torch.float8_e4m3fn).float() * xs.cpu()[:, None]).double()
    Wp, swd = pack_tiles_fp8(qw.to(dev)), torch.cat([sw, torch.ones((-N) % 16)]).to(dev)
    for nsplit in (1, 3):
        out = torch.zeros(nsplit, M, N, device=dev)
        lib.call("vr2_gemv_a8w8", q8, K, xs, M, Wp, swd, N, K, out, out.stride(0) if nsplit > 1 else 0, N, nsplit, S())
        got = out.sum(0).cpu().double()
        want = xd @ Wd.t()
        assert (got - want).abs().max() <= 5e-5 * want.abs().max(), (nsplit, (got - want).abs().max())   # fp32 sums of 1152 exact products, two orders
    # gate/up + SiLU*mul: I = 208 output columns from 2 x 208 interleaved weight rows
    I = 208
    Wg, Wu = torch.randn(I, K, generator=g) * 0.05, torch.randn(I, K, generator=g) * 0.05
    qg, sg = quantize_rows_fp8(Wg)
    qu, su = quantize_rows_fp8(Wu)
    qi, si = interleave_gate_up(qg.to(dev), qu.to(dev)), interleave_gate_up(sg.to(dev), su.to(dev)).contiguous()
    act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemv_gateup_silu_a8w8", q8, K, xs, M, pack_tiles_fp8(qi), si, I, K, act, I, S())
    gd, ud = xd @ dequantize_rows_fp8(qg, sg).double().t(), xd @ dequantize_rows_fp8(qu, su).double().t()
    want = (torch.nn.functional.silu(gd) * ud)
    assert rel_err(act.float().cpu().numpy(), want.numpy(), floor=want.abs().max().item() * 1e-2).max() < 1e-2   # one bf16 rounding
    # exact integers: e4m3 holds -8..8 exactly, power-of-two scales, every partial sum an integer below 2^24
    Wi = torch.randint(-8, 9, (N, K), generator=g).float() * 0.25
    xi = torch.randint(-4, 5, (M, K), generator=g).float()
    xi[:, 0] = 4.0                                                        # amax 4 -> scale 4/448 = 2^-7 * ... (not a power of two: check via dequant)
    qwi, swi = quantize_rows_fp8(Wi)
    lib.call("vr2_quant_rows_fp8", xi.to(torch.bfloat16).to(dev), K, M, K, q8, K, xs, S())
    out = torch.zeros(1, M, N, device=dev)
    lib.call("vr2_gemv_a8w8", q8, K, xs, M, pack_tiles_fp8(qwi.to(dev)), torch.cat([swi, torch.ones((-N) % 16)]).to(dev), N, K, out, 0, N, 1, S())
    xq = (_deinterleave_fp8_rows(q8.cpu(), K).view(torch.float8_e4m3fn).float() * xs.cpu()[:, None]).double()
    want = xq @ dequantize_rows_fp8(qwi, swi).double().t()
    assert (out[0].cpu().double() - want).abs().max() <= 1e-6 * want.abs().max()   # integer sums x two fp32 scale factors


def test_add_rmsnorm_with_fused_fp8_quantisation(dev):
    """vr2_add_rmsnorm_q8 = vr2_add_rmsnorm followed by vr2_quant_rows_fp8 on its bf16 output, byte for byte (split-K slabs + bias)"""
    M, H, ns = 37, 3584, 3
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(M, H, generator=g)
    part = (0.3 * torch.randn(ns, M, H, generator=g)).to(dev)
    bias = (0.1 * torch.randn(H, generator=g)).to(dev)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    xa, xb = x0.clone().to(dev), x0.clone().to(dev)
    ya = torch.zeros(M, H, dtype=torch.bfloat16, device=dev)
    yb = torch.zeros_like(ya)
    lib.call("vr2_add_rmsnorm", xa, H, part, part.stride(0), part.stride(1), ns, bias, w, ya, H, M, H, 1e-6, S())
    qa, sa = torch.zeros((M, H), dtype=torch.uint8, device=dev), torch.zeros(M, device=dev)
    lib.call("vr2_quant_rows_fp8", ya, H, M, H, qa, H, sa, S())
    qb, sb = torch.full((M, H + 128), 7, dtype=torch.uint8, device=dev), torch.zeros(M, device=dev)
    lib.call("vr2_add_rmsnorm_q8", xb, H, part, part.stride(0), part.stride(1), ns, bias, w, yb, H, M, H, 1e-6, qb, qb.stride(0), sb, S())
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and torch.equal(ya, yb) and torch.equal(sa, sb)
    assert torch.equal(qa, qb[:, :H]) and bool((qb[:, H:] == 7).all())


@pytest.mark.parametrize("M,N,K", [(40, 424, 1152),        # 64-row tiles, ragged N
                                   (1088, 4608, 3584),     # 128-row tiles (prefill of one prompt)
                                   (2304, 4608, 1152),     # 256-row tiles, one workgroup per CU
                                   (2368, 3584, 2048)])    # 256-row main + 64 tail rows as k-parts (atomic adds)
def test_gemm_fp8_operands_on_the_scaled_mfma(dev, M, N, K):
    """vr2_gemm_a8w8 (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales): per-row e4m3 activations x per-channel e4m3 weights
    against the float64 matmul of the DEQUANTISED operands -- products of two e4m3 values are exact in fp32, only the summation
    order differs -- for every tiling the launcher picks and the F32 / ADD_F32 / BF16 / SwiGLU epilogues; plus an exact-integer
    case where every sum is exact (any fragment-order mistake between the two operands is a hard mismatch)"""
    from video_r2_amd.weights import dequantize_rows_fp8, interleave_gate_up, pack_tiles_fp8, quantize_rows_fp8
    g = torch.Generator().manual_seed(7 * M + N)
    W = torch.randn(N, K, generator=g) * 0.05 * torch.logspace(-1, 1, N)[:, None]
    x = (2 * torch.randn(M, K, generator=g) * torch.logspace(-1, 1, M)[:, None]).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    qw, sw = quantize_rows_fp8(W)
    Wd = dequantize_rows_fp8(qw, sw).double()
    ldq = K + 128                                                          # a row stride above K
    q8 = torch.zeros((M, ldq), dtype=torch.uint8, device=dev)
    xs = torch.zeros(M, device=dev)
    lib.call("vr2_quant_rows_fp8", x.to(dev), K, M, K, q8, ldq, xs, S())
    xd = (_deinterleave_fp8_rows(q8[:, :K].cpu(), K).view(torch.float8_e4m3fn).float() * xs.cpu()[:, None]).double()
    Wp, swd = pack_tiles_fp8(qw.to(dev)), torch.cat([sw, torch.ones((-N) % 16)]).to(dev)
    want = xd @ Wd.t() + bias.double()
    tol = 5e-5 * want.abs().max()
    out = torch.full((M, N), 3.0, device=dev)
    lib.call("vr2_gemm_a8w8", q8, ldq, xs, Wp, swd, bias.to(dev), out, N, M, N, K, 1, S())        # F32
    assert (out.cpu().double() - want).abs().max() <= tol, (out.cpu().double() - want).abs().max()
    res = torch.randn(M, N, generator=g)
    out2 = res.clone().to(dev)
    lib.call("vr2_gemm_a8w8", q8, ldq, xs, Wp, swd, bias.to(dev), out2, N, M, N, K, 2, S())       # ADD_F32
    assert (out2.cpu().double() - (want + res.double())).abs().max() <= tol
    out3 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_a8w8", q8, ldq, xs, Wp, swd, None, out3, N, M, N, K, 0, S())               # BF16, no bias
    w3 = (xd @ Wd.t())
    assert rel_err(out3.float().cpu().numpy(), w3.numpy(), floor=w3.abs().max().item() * 1e-2).max() < 1e-2
    # gate/up + SiLU*mul on interleaved weight rows
    I = (N // 2) // 16 * 16
    Wg, Wu = W[:I], W[I:2 * I]
    qg, sg = quantize_rows_fp8(Wg)
    qu, su = quantize_rows_fp8(Wu)
    qi, si = interleave_gate_up(qg.to(dev), qu.to(dev)), interleave_gate_up(sg.to(dev), su.to(dev)).contiguous()
    act = torch.zeros(M, I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_gemm_a8w8", q8, ldq, xs, pack_tiles_fp8(qi), si, None, act, I, M, 2 * I, K, 4, S())
    gd, ud = xd @ dequantize_rows_fp8(qg, sg).double().t(), xd @ dequantize_rows_fp8(qu, su).double().t()
    wa = torch.nn.functional.silu(gd) * ud
    assert rel_err(act.float().cpu().numpy(), wa.numpy(), floor=wa.abs().max().item() * 1e-2).max() < 1e-2
    # exact integers
    Wi = torch.randint(-8, 9, (N, K), generator=g).float() * 0.25
    xi = torch.randint(-4, 5, (M, K), generator=g).float()
    xi[:, 0] = 4.0
    qwi, swi = quantize_rows_fp8(Wi)
    lib.call("vr2_quant_rows_fp8", xi.to(torch.bfloat16).to(dev), K, M, K, q8, ldq, xs, S())
    lib.call("vr2_gemm_a8w8", q8, ldq, xs, pack_tiles_fp8(qwi.to(dev)), torch.cat([swi, torch.ones((-N) % 16)]).to(dev), None, out, N, M, N, K, 1, S())
    xq = (_deinterleave_fp8_rows(q8[:, :K].cpu(), K).view(torch.float8_e4m3fn).float() * xs.cpu()[:, None]).double()
    wi = xq @ dequantize_rows_fp8(qwi, swi).double().t()
    assert (out.cpu().double() - wi).abs().max() <= 2e-6 * wi.abs().max()


@pytest.mark.parametrize("T,k,top_p,min_p", [(0.9, 50, 0.9, 0.0), (0.7, 50, 0.5, 0.05), (1.3, 20, 1.0, 0.2), (0.9, 64, 0.99, 0.01)])
def test_sample_top_p_and_min_p_match_the_hf_warpers(dev, T, k, top_p, min_p):
    """vr2_sample_topk_p: the support of the draws is EXACTLY the set HF's temperature -> top-k -> top-p -> min-p warpers keep
    (oracle.model.warp_logits, itself pinned against the HF classes on CPU) and the frequencies follow the renormalised
    distribution (chi-square); top_p = 1, min_p = 0 reproduces vr2_sample_topk draw for draw; requests the two-stage path cannot
    serve are refused."""
    from oracle import model as om
    V, G = 4096, 8
    ws = torch.zeros(int(lib.load().vr2_sample_ws_bytes(G)) // 4, device=dev)
    g = torch.Generator().manual_seed(5)
    logits = (3.0 * torch.randn(1, V, generator=g))
    kept = om.warp_logits(logits, T, k, top_p, min_p or None)[0]
    keep = ~torch.isinf(kept)
    p = kept.softmax(-1)
    ld = logits.to(dev)
    nxt = torch.zeros(G, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    counts = torch.zeros(V)
    for it in range(600):
        lib.call("vr2_sample_topk_p", ld, 0, G, V, T, k, top_p, min_p, 4321 + it, None, step, None, -1, 0, nxt, None, 0, ws, S())
        counts += torch.bincount(nxt.cpu().long(), minlength=V).float()
    assert counts[~keep].sum().item() == 0, "a token outside the warped set was drawn"
    n = counts.sum().item()
    exp = p * n
    assert bool((counts[exp > 20] > 0).all()), "a kept token with sizeable probability never appeared"
    sel = exp > 5
    if int(sel.sum()) > 1:
        chi2 = (((counts - exp) ** 2)[sel] / exp[sel]).sum().item()
        dof = int(sel.sum().item()) - 1
        assert chi2 < dof + 6 * (2 * dof) ** 0.5, (chi2, dof)
    a = torch.zeros(G, dtype=torch.int32, device=dev); b = torch.zeros(G, dtype=torch.int32, device=dev)
    lib.call("vr2_sample_topk_p", ld, 0, G, V, T, k, 1.0, 0.0, 77, None, step, None, -1, 0, a, None, 0, ws, S())
    lib.call("vr2_sample_topk", ld, 0, G, V, T, k, 77, None, step, None, -1, 0, b, None, 0, ws, S())
    assert torch.equal(a, b)
    L = lib.load()
    assert L.vr2_sample_topk_p(ld.data_ptr(), 0, G, V, T, 200, 0.9, 0.0, 1, None, step.data_ptr(), None, -1, 0, nxt.data_ptr(), None, 0, ws.data_ptr(), None) != 0
    assert b"top_k" in L.vr2_last_error()


def test_sample_warpers_are_reproducible_under_heavy_ties(dev):
    """long runs of EQUAL logits: more than CAND (64) values tied at a slice's top-k threshold and more than 512 kept scores in the
    warper list.  The candidate slots and the list slots are assigned in token-index order (no atomic counters), so which tied tokens
    survive (the lowest indices) and the order of the sums over the kept set do not depend on scheduling: a fixed seed gives the
    same tokens on every run, with and without top_p / min_p."""
    V, G = 32 * 2048, 16
    ws = torch.zeros(int(lib.load().vr2_sample_ws_bytes(G)) // 4, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(9)
    flat = torch.zeros(1, V)                                             # every token tied
    plateau = torch.randn(1, V, generator=g)
    plateau[0, 100:1300] = 7.0                                           # 1200 tokens tied at the top inside ONE slice (slice = 2048)
    few = torch.randn(1, V, generator=g)
    few[0, 5000:5040] = 6.0                                              # 40 tied winners: within every capacity
    for logits in (flat, plateau, few):
        ld = logits.to(dev)
        for (k, top_p, min_p) in ((50, 1.0, 0.0), (50, 0.7, 0.0), (64, 0.9, 0.05)):
            draws = []
            for rep in range(6):
                nxt = torch.full((G,), -1, dtype=torch.int32, device=dev)
                lib.call("vr2_sample_topk_p", ld, 0, G, V, 1.0, k, top_p, min_p, 1234, None, step, None, -1, 0, nxt, None, 0, ws, S())
                draws.append(nxt.cpu().clone())
            assert all(torch.equal(draws[0], d) for d in draws[1:]), (k, top_p, min_p)
            assert int(draws[0].min()) >= 0 and int(draws[0].max()) < V
        # the tied winners that survive a full candidate list are the LOWEST indices of the plateau
        if logits is plateau:
            nxt = torch.zeros(G, dtype=torch.int32, device=dev)
            seen = set()
            for it in range(200):
                lib.call("vr2_sample_topk_p", ld, 0, G, V, 1.0, 50, 1.0, 0.0, 99 + it, None, step, None, -1, 0, nxt, None, 0, ws, S())
                seen.update(nxt.cpu().tolist())
            assert seen and min(seen) >= 100 and max(seen) < 100 + 64, (min(seen), max(seen))


def test_repetition_penalty_matches_hf_and_tracks_new_tokens(dev):
    """vr2_repetition_penalty against oracle.model.repetition_penalty (pinned against HF's RepetitionPenaltyLogitsProcessor): ids
    seen so far are penalised ONCE whatever their multiplicity, the token appended since the last call is folded into the bitmap"""
    from oracle import model as om
    G, V = 5, 3001
    g = torch.Generator().manual_seed(8)
    x = 2.0 * torch.randn(G, V, generator=g)
    prompt = torch.randint(0, V, (G, 30), generator=g)
    prompt[:, 5] = prompt[:, 4]                                   # a repeated id
    words = (V + 31) // 32
    bits = np.zeros((G, words * 32), dtype=bool)
    for r in range(G):
        bits[r, prompt[r].numpy()] = True
    seen = torch.from_numpy(np.packbits(bits, axis=1, bitorder="little").view(np.int32).copy()).to(dev)
    xd = x.clone().to(dev)
    lib.call("vr2_repetition_penalty", xd, V, G, V, 1.3, seen, words, None, S())
    assert torch.equal(xd.cpu(), om.repetition_penalty(x, prompt, 1.3))
    new = torch.tensor([7, 3000, int(prompt[2, 0]), 64, 31], dtype=torch.int32)
    x2 = 2.0 * torch.randn(G, V, generator=g)
    xd2 = x2.clone().to(dev)
    lib.call("vr2_repetition_penalty", xd2, V, G, V, 0.8, seen, words, new.to(dev), S())
    hist = torch.cat([prompt, new.long()[:, None]], 1)
    assert torch.equal(xd2.cpu(), om.repetition_penalty(x2, hist, 0.8))
    x3 = 2.0 * torch.randn(G, V, generator=g)
    xd3 = x3.clone().to(dev)
    lib.call("vr2_repetition_penalty", xd3, V, G, V, 1.1, seen, words, None, S())      # the new ids stayed in the bitmap
    assert torch.equal(xd3.cpu(), om.repetition_penalty(x3, hist, 1.1))
