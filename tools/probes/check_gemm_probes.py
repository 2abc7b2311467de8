"""GPU: the persistent bf16 GEMM of the encoders' Linear layers (csrc/gemm.hip) against an f32 torch reference of the same
op (F.linear on the bf16-rounded operands), incl. ragged edges, bias, activations and the XCD tile order at odd grids."""

import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _ref(a, b, bias, act):
    y = a.float() @ b.float().t()
    if bias is not None:
        y = y + bias
    pre = y
    if act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    return y, pre


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 768), (4096, 2304, 768), (1000, 520, 128), (8192 + 77, 3072, 768),
                                   (25088, 384, 1024), (2048, 768, 3072), (300, 8, 64)])
@pytest.mark.parametrize("out", [torch.bfloat16, torch.float32])
def test_gemm_nt_vs_f32_reference(M, N, K, out):
    import gemm_probe as GP

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    for use_bias in (False, True):
        c = GP.gemm_nt(a, b, bias if use_bias else None, None, out)
        ref, _ = _ref(a, b, bias if use_bias else None, None)
        tol = 1e-2 if out == torch.bfloat16 else 1e-4
        err = (c.float() - ref).abs().max().item()
        assert err <= tol * ref.abs().max().item(), (M, N, K, use_bias, err)
    # row-strided operands (views into wider buffers) and a repeat on the same stream (ring state restarts cleanly)
    wide = torch.randn(M, K + 64, generator=g).to(dev).bfloat16()
    c = GP.gemm_nt(wide[:, :K], b, None, None, out)
    ref, _ = _ref(wide[:, :K], b, None, None)
    assert (c.float() - ref).abs().max().item() <= (1e-2 if out == torch.bfloat16 else 1e-4) * ref.abs().max().item()


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_gemm_nt_bias_activation_epilogue(act):
    import gemm_probe as GP

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 1500, 3072, 768
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    y, pre = GP.gemm_nt(a, b, bias, act, torch.bfloat16, want_pre=True)
    ref, ref_pre = _ref(a, b, bias, act)
    assert (pre.float() - ref_pre).abs().max().item() <= 1e-2 * ref_pre.abs().max().item()
    assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    y2 = GP.gemm_nt(a, b, bias, act, torch.bfloat16)
    assert torch.equal(y2, y)


def test_gemm_nt_is_deterministic_and_independent_of_the_grid(monkeypatch):
    import gemm_probe as GP

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(9)
    a = torch.randn(20000, 768, generator=g).to(dev).bfloat16()
    b = (torch.randn(2304, 768, generator=g) / 28).to(dev).bfloat16()
    c0 = GP.gemm_nt(a, b)
    for _ in range(3):
        assert torch.equal(GP.gemm_nt(a, b), c0)
    ref = a.float() @ b.float().t()
    assert (c0.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


def test_unsupported_shapes_are_refused():
    import gemm_probe as GP

    assert not GP.gemm_nt_supported(100, 768, 768, 768, 768, 768)      # fewer than one tile of rows
    assert not GP.gemm_nt_supported(4096, 768, 100, 100, 100, 768)     # K not a multiple of 64
    assert GP.gemm_nt_supported(201728, 768, 768, 768, 768, 768)


# ------------------------------------------------------------------ four-wave design (csrc/gemm4.hip)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 768), (4096, 2304, 768), (8192, 3072, 768), (25088, 1024, 1024),
                                   (2048, 768, 3072), (78848, 768, 768)])
@pytest.mark.parametrize("out", [torch.bfloat16, torch.float32])
def test_gemm4_nt_vs_f32_reference(M, N, K, out):
    """One to many tiles per workgroup, K steps from 1 (ring barely started) to 48, both output types, bias, row-strided A."""
    import gemm_probe as GP

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    tol = 1e-2 if out == torch.bfloat16 else 1e-4
    for use_bias in (False, True):
        c = GP.gemm4_nt(a, b, bias if use_bias else None, None, out)
        ref, _ = _ref(a, b, bias if use_bias else None, None)
        err = (c.float() - ref).abs().max().item()
        assert err <= tol * ref.abs().max().item(), (M, N, K, use_bias, err)
    if M <= 8192:
        wide = torch.randn(M, K + 64, generator=g).to(dev).bfloat16()
        c = GP.gemm4_nt(wide[:, :K], b, None, None, out)
        ref, _ = _ref(wide[:, :K], b, None, None)
        assert (c.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
        assert torch.equal(GP.gemm4_nt(wide[:, :K], b, None, None, out), c)   # deterministic, ring state restarts cleanly


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_gemm4_nt_bias_activation_epilogue(act):
    import gemm_probe as GP

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 1536, 3072, 768
    a = torch.randn(M, K, generator=g).to(dev).bfloat16()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    y, pre = GP.gemm4_nt(a, b, bias, act, torch.bfloat16, want_pre=True)
    ref, ref_pre = _ref(a, b, bias, act)
    assert (pre.float() - ref_pre).abs().max().item() <= 1e-2 * ref_pre.abs().max().item()
    assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    assert torch.equal(GP.gemm4_nt(a, b, bias, act, torch.bfloat16), y)
    # same k order and f32 accumulation as the eight-wave kernel: bit-identical outputs
    assert torch.equal(GP.gemm_nt(a, b, bias, act, torch.bfloat16), y)
