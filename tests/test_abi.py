"""CPU: the C-ABI library builds, loads, and exports every symbol include/vr2.h declares (no compute calls)."""
import ctypes
import os

from video_r2_amd import build, lib


def test_build_and_symbols():
    path = build.build()
    assert os.path.exists(path)
    protos = lib.parse_header()
    assert len(protos) >= 25
    dll = ctypes.CDLL(path)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/vr2.h but not exported by libvr2.so"


def test_loader_binds_all_and_reports_version():
    l = lib.load()
    assert lib.version() == 100
    assert l.vr2_last_error() is not None


def test_pack_tiles_roundtrip_and_fragment_order():
    import torch
    from video_r2_amd.weights import pack_tiles, unpack_tiles
    w = torch.arange(40 * 64, dtype=torch.float32).reshape(40, 64).to(torch.bfloat16)
    p = pack_tiles(w)
    assert p.shape == (3, 2 * 512)                      # 40 rows -> 3 tiles of 16 (zero padded), 64 k -> 2 k-steps
    assert torch.equal(unpack_tiles(p, 40, 64), w)
    # element (n, k) sits in tile (n//16, k//32) at slot 16*((k%32)//8) + n%16, position k%8 (include/vr2.h)
    for n, k in [(0, 0), (5, 9), (17, 40), (39, 63), (16, 31)]:
        off = (k // 32) * 512 + (16 * ((k % 32) // 8) + n % 16) * 8 + k % 8
        assert p[n // 16, off] == w[n, k]
    assert (p[2].view(2, 4, 16, 8)[:, :, 8:] == 0).all()  # padded rows are zero


def test_argument_validation_without_gpu():
    # shape validation happens on the host before any launch: bad K must be refused with a message, not a crash
    l = lib.load()
    rc = l.vr2_gemm_bf16(None, 8, None, None, None, 8, 4, 8, 100, 0, None)
    assert rc != 0 and b"multiple of 64" in l.vr2_last_error()
    rc = l.vr2_gemv_bf16(None, 8, 257, None, 16, 128, None, 0, 16, 1, None)
    assert rc != 0 and b"[1,256]" in l.vr2_last_error()
    rc = l.vr2_gemv_bf16(None, 8, 129, None, 16, 128, None, 0, 16, 1, None)      # above 128 rows only the ring kernel runs: whole 64-k chunks, >= 4 per part
    assert rc != 0 and b"> 128 rows needs" in l.vr2_last_error()
    rc = l.vr2_attn_tok(None, 1, None, 8, None, None, None, None, 8, None, 8, 4, 2, 64, 1.0, None, None)
    assert rc != 0 and b"head_dim" in l.vr2_last_error()


def test_fp8_tile_layout_and_quantiser():
    import torch
    """weights.pack_tiles_fp8: tile (n/16, k/64) of 1 KiB; lane l = 16*q + n%16 owns 16 bytes: k = 64*kt + 8q .. +7 (first MFMA
    k-step) then k = 64*kt + 32 + 8q .. +7 (include/vr2.h, csrc/gemv_w8.hip); quantise -> dequantise stays at the e4m3 level"""
    from video_r2_amd.weights import dequantize_rows_fp8, pack_tiles_fp8, quantize_rows_fp8
    n, k = 40, 192
    q8 = (torch.arange(n * k, dtype=torch.int64) % 251).to(torch.uint8).view(n, k)
    p = pack_tiles_fp8(q8)
    assert p.shape == (3, (k // 64) * 1024) and p.dtype == torch.uint8
    for row, col in [(0, 0), (5, 9), (17, 40), (39, 63), (16, 31), (33, 64 + 37), (2, 191)]:
        kt, kin = col // 64, col % 64
        half, q, e = kin // 32, (kin % 32) // 8, kin % 8
        off = kt * 1024 + (16 * q + row % 16) * 16 + half * 8 + e
        assert p[row // 16, off] == q8[row, col], (row, col)
    assert (p[2].view(k // 64, 4, 16, 16)[:, :, 8:] == 0).all()          # padded rows 40..47 are zero
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 128, generator=g) * torch.logspace(-3, 1, 24)[:, None]
    w[7] = 0
    q, s = quantize_rows_fp8(w)
    assert q.dtype == torch.uint8 and s.shape == (24,) and s[7] == 1.0
    d = dequantize_rows_fp8(q, s)
    assert torch.equal(d[7], torch.zeros(128))
    rel = (d - w).abs().amax(1) / w.abs().amax(1).clamp(min=1e-30)
    assert rel.max() <= 2 ** -4                                             # 3 mantissa bits, per-row scale to 448
    assert (dequantize_rows_fp8(*quantize_rows_fp8(d)) - d).abs().max() <= 1e-6 * d.abs().max()   # idempotent on its own output
