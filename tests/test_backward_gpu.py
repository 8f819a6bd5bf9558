"""GPU parity of the update-path kernels (backward + optimiser) through the C ABI against torch-CPU autograd / torch.optim.AdamW."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from video_r2_amd import lib  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    lib.load()
    return torch.device("cuda:0")


def S():
    return torch.cuda.current_stream().cuda_stream


def bf(x):
    return x.to(torch.bfloat16)


def test_gemm_tn_exact_and_random(dev):
    g = torch.Generator().manual_seed(1)
    K, M, N = 200, 136, 264                                   # K (tokens) ragged vs the 64-row k-tile
    A = torch.randint(-3, 4, (K, M), generator=g).float(); B = torch.randint(-2, 3, (K, N), generator=g).float()
    B[:, ::3] += 1
    C = torch.full((M, N), 5.0, device=dev)
    lib.call("vr2_gemm_tn_bf16", bf(A).to(dev), M, bf(B).to(dev), N, C, N, M, N, K, 0, S())
    assert torch.equal(C.cpu(), A.t() @ B)                     # exact: any fragment-layout / k-permutation error is a hard mismatch
    lib.call("vr2_gemm_tn_bf16", bf(A).to(dev), M, bf(B).to(dev), N, C, N, M, N, K, 1, S())
    assert torch.equal(C.cpu(), 2 * (A.t() @ B))
    K, M, N = 1000, 512, 1152
    A = bf(torch.randn(K, M, generator=g)); B = bf(torch.randn(K, N, generator=g))
    C = torch.zeros(M, N, device=dev)
    lib.call("vr2_gemm_tn_bf16", A.to(dev), M, B.to(dev), N, C, N, M, N, K, 0, S())
    ref = A.float().t() @ B.float()
    assert (C.cpu() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("K", [64, 128, 200, 456])
def test_gemm_tn_256_tile_exact(dev, K):
    """weight-gradient shapes the launcher routes to the 256 x 256 tile (8 waves, transposing LDS reads, counted-vmcnt pipeline):
    ragged feature edges on both operands, 1-7 k-tiles, a k tail (K % 64 != 0) finished by the 128-tile kernel"""
    M, N = 4000, 3592                                          # 16 x 15 tiles of 256 = 240 workgroups (one round)
    tm, tn = -(-M // 256), -(-N // 256)
    assert tm * tn * 4 >= 3 * -(-tm * tn // 256) * 256
    g = torch.Generator().manual_seed(K)
    A = torch.randint(-3, 4, (K, M), generator=g).float(); B = torch.randint(-2, 3, (K, N), generator=g).float()
    B[:, ::3] += 1
    ref = A.t() @ B
    C = torch.full((M + 1, N), 5.0, device=dev)
    lib.call("vr2_gemm_tn_bf16", bf(A).to(dev), M, bf(B).to(dev), N, C, N, M, N, K, 0, S())
    assert torch.equal(C[:M].cpu(), ref) and (C[M] == 5.0).all()
    lib.call("vr2_gemm_tn_bf16", bf(A).to(dev), M, bf(B).to(dev), N, C, N, M, N, K, 1, S())
    assert torch.equal(C[:M].cpu(), 2 * ref) and (C[M] == 5.0).all()


def test_gemm_tn_partial_last_round_splits_k(dev):
    """1036 tiles of 256 x 256 = 4.05 rounds on 256 CUs: the (+)= launch runs 1024 tiles as they are and the last 12 as
    tiles x k-parts with atomic adds (gemm_tn.hip); exact on small integers whatever the order of the adds"""
    M, N, K = 3584, 18944, 1088                                  # 14 x 74 tiles; 17 k-tiles -> 2 k-parts
    g = torch.Generator(device=dev).manual_seed(3)
    A = torch.randint(-3, 4, (K, M), generator=g, device=dev).float()
    B = torch.randint(-2, 3, (K, N), generator=g, device=dev).float()
    ref = A.t() @ B
    C = torch.full((M + 1, N), 5.0, device=dev)
    lib.call("vr2_gemm_tn_bf16", A.to(torch.bfloat16), M, B.to(torch.bfloat16), N, C, N, M, N, K, 1, S())
    assert torch.equal(C[:M], ref + 5.0) and (C[M] == 5.0).all()
    lib.call("vr2_gemm_tn_bf16", A.to(torch.bfloat16), M, B.to(torch.bfloat16), N, C, N, M, N, K, 0, S())
    assert torch.equal(C[:M], ref)


def test_swiglu_fwd_bwd(dev):
    from video_r2_amd.weights import interleave_gate_up
    T, I = 37, 48
    g = torch.Generator().manual_seed(2)
    gate = torch.randn(T, I, generator=g, requires_grad=True); up = torch.randn(T, I, generator=g, requires_grad=True)
    gq, uq = bf(gate.detach()).float().requires_grad_(), bf(up.detach()).float().requires_grad_()
    act = F.silu(gq) * uq
    dact = bf(torch.randn(T, I, generator=g)).float()
    act.backward(dact)
    gu = interleave_gate_up(bf(gq.detach()).t().contiguous(), bf(uq.detach()).t().contiguous()).t().contiguous()   # [T][2*I]
    out = torch.zeros(T, I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_swiglu_fwd", gu.to(dev), 2 * I, out, I, T, I, S())
    assert (out.float().cpu() - act.detach()).abs().max().item() <= 2 ** -8 * act.abs().max().item()
    dgu = torch.zeros(T, 2 * I, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_swiglu_bwd", gu.to(dev), 2 * I, bf(dact).to(dev), I, dgu, 2 * I, T, I, S())
    d = dgu.float().cpu().view(T, I // 16, 2, 16)
    assert (d[:, :, 0].reshape(T, I) - gq.grad).abs().max().item() <= 2 ** -8 * gq.grad.abs().max().item()
    assert (d[:, :, 1].reshape(T, I) - uq.grad).abs().max().item() <= 2 ** -8 * uq.grad.abs().max().item()


@pytest.mark.parametrize("M,H", [(300, 512), (1030, 3584), (4100, 2048), (70, 8192)])
def test_rmsnorm_bwd(dev, M, H):
    """one-pass kernel (rows + weight gradient, 1 / 4 / 8 rows per workgroup) for H <= 4096; two passes above"""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, H, generator=g, requires_grad=True); w = (1 + 0.1 * torch.randn(H, generator=g)).requires_grad_()
    dy = torch.randn(M, H, generator=g)
    y = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    y.backward(dy)
    dx = torch.full((M, H), 1.0, device=dev); rinv = torch.zeros(M, device=dev); dw = torch.zeros(H, device=dev)
    dxb = torch.zeros(M, H + 8, dtype=torch.bfloat16, device=dev)   # optional bf16 copy of the final dx rows, own row stride
    lib.call("vr2_rmsnorm_bwd", x.detach().to(dev), H, w.detach().to(dev), dy.to(dev), H, dx, H, rinv, dw, M, H, 1e-6, 1, dxb, H + 8, S())
    assert (dx.cpu() - 1.0 - x.grad).abs().max().item() < 1e-5 * max(1.0, x.grad.abs().max().item())
    assert torch.equal(dxb[:, :H], dx.to(torch.bfloat16)) and (dxb[:, H:] == 0).all()
    assert (dw.cpu() - w.grad).abs().max().item() < 1e-4 * w.grad.abs().max().item()
    dx2 = torch.full((M, H), 1.0, device=dev)                      # no weight gradient, overwrite
    lib.call("vr2_rmsnorm_bwd", x.detach().to(dev), H, w.detach().to(dev), dy.to(dev), H, dx2, H, None, None, M, H, 1e-6, 0, None, 0, S())
    assert (dx2.cpu() - x.grad).abs().max().item() < 1e-5 * max(1.0, x.grad.abs().max().item())


def test_rope_bwd_is_transpose_of_forward(dev):
    hd, nq, nkv, T = 128, 4, 2, 19
    g = torch.Generator().manual_seed(4)
    ang = torch.rand(T, hd // 2, generator=g) * 6
    cs, sn = ang.cos().to(dev), ang.sin().to(dev)
    x = torch.randn(T, (nq + 2 * nkv) * hd, generator=g)
    dy = bf(torch.randn(T, (nq + 2 * nkv) * hd, generator=g))
    qo = torch.zeros(T, nq * hd, dtype=torch.bfloat16, device=dev); ko = torch.zeros(T, nkv * hd, dtype=torch.bfloat16, device=dev)
    vo = torch.zeros_like(ko)
    lib.call("vr2_rope_qkv", x.to(dev), x.stride(0), cs, sn, T, nq, nkv, hd, qo, nq * hd, ko, vo, nkv * hd, 0, S())
    dq, dk, dv = dy[:, :nq * hd].contiguous(), dy[:, nq * hd:(nq + nkv) * hd].contiguous(), dy[:, (nq + nkv) * hd:].contiguous()
    dx = torch.zeros(T, (nq + 2 * nkv) * hd, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_rope_bwd", dq.to(dev), nq * hd, dk.float().to(dev), dv.float().to(dev), nkv * hd, cs, sn, T, nq, nkv, hd, dx, dx.stride(0), S())
    # <rope(x), dy> == <x, rope_bwd(dy)> (adjoint identity), up to the bf16 roundings of both outputs
    fwd = torch.cat([qo, ko, vo], 1).float().cpu()
    lhs = (fwd * dy.float()).sum().item(); rhs = (x * dx.float().cpu()).sum().item()
    assert abs(lhs - rhs) < 5e-3 * max(abs(lhs), (x.norm() * dy.float().norm()).item() * 0.01)


def test_dlogits_embed_bwd_colsum(dev):
    R, V = 21, 1000
    g = torch.Generator().manual_seed(5)
    z = (3 * torch.randn(R, V, generator=g)).requires_grad_()
    tok = torch.randint(0, V, (R,), generator=g); dlp = torch.randn(R, generator=g)
    lp = torch.gather(z / 0.9, 1, tok[:, None])[:, 0] - torch.logsumexp(z / 0.9, -1)
    lp.backward(dlp)
    lse = torch.logsumexp(z.detach() / 0.9, -1)
    out = torch.full((R, 1024), 9.0, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_dlogits", z.detach().to(dev), V, R, V, tok.to(torch.int32).to(dev), lse.to(dev), dlp.to(dev), 0.9, out, 1024, S())
    assert (out.float().cpu()[:, :V] - z.grad).abs().max().item() <= 2 ** -8 * z.grad.abs().max().item() + 1e-6
    assert (out[:, V:] == 0).all()
    T, H, NV = 40, 64, 50
    ids = torch.randint(0, NV, (T,), generator=g); ids[3] = 7; ids[9] = 7
    dx = torch.randn(T, H, generator=g)
    tab = torch.zeros(NV, H, device=dev)
    skip = torch.tensor([7], dtype=torch.int32, device=dev)
    lib.call("vr2_embed_bwd", ids.to(torch.int32).to(dev), skip, dx.to(dev), H, tab, H, T, H, S())
    ref = torch.zeros(NV, H); ref.index_add_(0, ids, dx); ref[7] = 0
    assert (tab.cpu() - ref).abs().max().item() < 1e-5
    a = bf(torch.randn(700, 96, generator=g)); cs = torch.zeros(96, device=dev)
    lib.call("vr2_colsum_bf16", a.to(dev), 96, cs, 700, 96, S())
    assert (cs.cpu() - a.float().sum(0)).abs().max().item() < 1e-3


@pytest.mark.parametrize("group,Sp,C,G,ktile,qsplit", [(2, 70, 40, 2, 128, 48), (7, 70, 40, 2, 64, 48), (7, 300, 200, 3, 128, 256),
                                                       (4, 256, 128, 2, 128, 128), (7, 1, 1, 1, 128, 512), (2, 129, 65, 2, 128, 64),
                                                       (1, 33, 3, 4, 128, 512)])
def test_attention_backward_vs_autograd(dev, group, Sp, C, G, ktile, qsplit):
    """Unified rows: S prompt rows (causal) + G sequences of C completion rows (prefix visible + own causal), GQA."""
    from video_r2_amd.host import vision_plan as vp
    hd, nKV = 128, 2
    nH = nKV * group
    T = Sp + G * C
    scale = hd ** -0.5
    g = torch.Generator().manual_seed(group)
    q = bf(torch.randn(T, nH, hd, generator=g)).float().requires_grad_()
    k = bf(torch.randn(T, nKV, hd, generator=g)).float().requires_grad_()
    v = bf(torch.randn(T, nKV, hd, generator=g)).float().requires_grad_()
    allow = torch.zeros(T, T, dtype=torch.bool)
    allow[:Sp, :Sp] = torch.tril(torch.ones(Sp, Sp, dtype=torch.bool))
    for gg in range(G):
        a = Sp + gg * C
        allow[a:a + C, :Sp] = True
        allow[a:a + C, a:a + C] = torch.tril(torch.ones(C, C, dtype=torch.bool))
    kr = k.repeat_interleave(group, 1); vr = v.repeat_interleave(group, 1)
    s = torch.einsum("qhd,khd->hqk", q, kr) * scale
    s = s.masked_fill(~allow[None], float("-inf"))
    p = s.softmax(-1)
    o = torch.einsum("hqk,khd->qhd", p, vr)
    do = bf(torch.randn(T, nH, hd, generator=g)).float()
    o.backward(do)
    lse_ref = torch.logsumexp(s.detach(), -1).t().contiguous()                 # [T][nH]
    # device: forward with LSE, then backward kernels
    qt = vp.LM_Q_TILE if Sp != 256 else vp.Q_TILE          # 64-row items stay valid for the 128-row kernels (half the waves idle)
    work = np.concatenate([vp.causal_work(Sp, 0, 0, 0, 0, q_tile=qt)] + [vp.causal_work(C, Sp + gg * C, 0, Sp, Sp + gg * C, q_tile=qt) for gg in range(G)])
    dwork = torch.from_numpy(work).to(dev)
    qd, kd, vd = bf(q.detach()).view(T, -1).to(dev), bf(k.detach()).view(T, -1).to(dev), bf(v.detach()).view(T, -1).to(dev)
    od = torch.zeros(T, nH * hd, dtype=torch.bfloat16, device=dev); lse = torch.zeros(T, nH, device=dev)
    lib.call("vr2_attn_tok", dwork, dwork.shape[0], qd, nH * hd, kd, vd, kd, vd, nKV * hd, od, nH * hd, nH, group, hd, scale, lse, S())
    assert (lse.cpu() - lse_ref).abs().max().item() < 2e-3
    dod = bf(do).view(T, -1).to(dev)
    rowc = torch.zeros(2, nH, T, device=dev)
    lib.call("vr2_attn_delta", od, dod, nH * hd, T, nH, hd, lse, rowc, S())
    delta_ref = (o.detach() * do).sum(-1)                                        # [T][nH]
    assert (rowc[1].t().cpu() - delta_ref).abs().max().item() < 2e-2 * delta_ref.abs().max().item()
    assert torch.allclose(rowc[0].t().cpu(), -lse.cpu() * 1.4426950408889634, rtol=1e-6, atol=1e-6)
    work_dq = np.concatenate([vp.causal_work(Sp, 0, 0, 0, 0, q_tile=128)] +
                             [vp.causal_work(C, Sp + gg * C, 0, Sp, Sp + gg * C, q_tile=128) for gg in range(G)])
    dwq = torch.from_numpy(work_dq).to(dev)
    dq = torch.zeros(T, nH * hd, dtype=torch.bfloat16, device=dev)
    lib.call("vr2_attn_bwd_dq", dwq, dwq.shape[0], qd, nH * hd, kd, vd, kd, vd, nKV * hd, dod, rowc, T, dq, nH, group, hd, scale, S())
    # small splits: several adders per prompt key row, by query range and by q heads
    kw = torch.from_numpy(vp.key_work(Sp, G, C, k_tile=ktile, q_split=qsplit, group=group, max_steps=3)).to(dev)
    assert group == 1 or Sp + C < 8 or (kw[:, 7] < group).any().item()     # some items cover only part of the GQA group
    dk = torch.zeros(T, nKV * hd, dtype=torch.float32, device=dev); dv = torch.zeros_like(dk)
    lib.call("vr2_attn_bwd_dkv", kw, kw.shape[0], qd, nH * hd, kd, vd, nKV * hd, dod, rowc, T, dk, dv, nH, group, hd, scale, S())
    # bf16 P / dS operands and bf16 outputs: 2^-6 of each gradient's scale
    for got, ref, name in ((dq, q.grad.view(T, -1), "dq"), (dk, k.grad.view(T, -1), "dk"), (dv, v.grad.view(T, -1), "dv")):
        e = (got.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert e < 2 ** -6, (name, e)


def test_adamw_matches_torch_and_repacks(dev):
    from video_r2_amd.weights import pack_tiles, unpack_tiles
    N, K = 40, 96
    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(N, K, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.clone().to(dev); m = torch.zeros(N, K, device=dev); v = torch.zeros(N, K, device=dev)
    wp = torch.zeros(3, 3 * 512 + 512, dtype=torch.bfloat16, device=dev)[:, :(128 // 32) * 512].contiguous()  # [N16/16][Kp/32*512], Kp=128
    wt = torch.zeros(96 // 16, (64 // 32) * 512, dtype=torch.bfloat16, device=dev)                                # W^T: [K/16][Np/32*512], Np=64
    rows = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
    gs = torch.tensor([0.5], device=dev)
    for step in range(1, 4):
        grad = torch.randn(N, K, generator=g)
        ref.grad = 0.5 * grad
        opt.step()
        lib.call("vr2_adamw", p, grad.to(dev), m, v, N, K, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, gs, wp, 128, wt, 64, rows, K, S())
        assert (p.cpu() - ref.detach()).abs().max().item() < 1e-6
    want = ref.detach().to(torch.bfloat16)
    assert torch.equal(rows.cpu(), want)
    assert torch.equal(unpack_tiles(wp.cpu(), N, 128)[:, :K], want)
    assert torch.equal(unpack_tiles(wt.cpu(), K, 64)[:, :N], want.t())
    # clipping coefficient
    # clipping coefficient: two tensors in two slots of per-block partials, summed without atomics -> bitwise reproducible
    slot = int(lib.load().vr2_sumsq_slot_floats())
    coef = torch.zeros(1, device=dev); nrm = torch.zeros(1, device=dev)
    gr = torch.randn(100000, generator=g); gr2 = torch.randn(3, generator=g)
    seen = set()
    for rep in range(3):
        ss = torch.full((2 * slot,), float("nan"), device=dev)
        lib.call("vr2_sumsq_f32", gr.to(dev), gr.numel(), ss, 0, S())
        lib.call("vr2_sumsq_f32", gr2.to(dev), gr2.numel(), ss, 1, S())
        lib.call("vr2_clip_coef", ss, 2, 1.0, 0.25, coef, nrm, S())
        seen.add((nrm.item(), coef.item()))
    assert len(seen) == 1
    n = torch.cat([gr, gr2]).norm().item() * 0.25
    assert abs(nrm.item() - n) < 1e-3 * n and abs(coef.item() - min(1.0, 1.0 / (n + 1e-6)) * 0.25) < 1e-6
