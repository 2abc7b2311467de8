"""GPU parity of the split-M weight-gradient GEMM (csrc/wgrad.hip) against a plain PyTorch f32 reference."""

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (1000, 256, 512), (77 * 13, 768, 3072), (64, 8, 8), (5000, 2304, 768),
                                   (130, 520, 264),
                                   # the 8-phase kernel (N, K multiples of 256, >= 4 K-tiles per split): ragged last K-tile of a split,
                                   # an odd number of K-tiles, run-ahead DMA past the split's end, the shortest split it accepts
                                   (4096 + 37, 256, 512), (64 * 9 + 1, 768, 256), (64 * 5, 256, 256), (2 * 8191, 512, 768), (12345, 1024, 256),
                                   # the 128 x 128-tile instantiation (narrow layers: HTSAT's 96 / 288 / 384-wide Linears, the I-JEPA
                                   # predictor's 384 / 1152 / 1536): full tiles, ragged tiles, M not a multiple of the 64-row stage,
                                   # more tiles than one XCD has workgroup slots (9 x 12 = 108 > 64)
                                   (8192, 96, 96), (8192, 288, 96), (8192, 384, 96), (8192, 96, 384), (4099, 384, 384), (6000, 1152, 384),
                                   (6000, 384, 1536), (1000, 104, 72), (129, 128, 128), (3000, 1152, 1536)])
def test_wgrad_vs_torch(M, N, K):
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).bfloat16()
    x = torch.randn(M, K, generator=g).bfloat16()
    ref = dy.float().t() @ x.float()
    for out_dtype in (torch.float32, torch.bfloat16):
        dw = Kn.wgrad(dy.to(dev), x.to(dev), out_dtype)
        assert dw.shape == (N, K) and dw.dtype == out_dtype
        err = (dw.float().cpu() - ref).abs().max().item()
        tol = (2e-3 if out_dtype == torch.float32 else 1e-2) * max(1.0, ref.abs().max().item())
        assert err <= tol, (out_dtype, err, ref.abs().max().item())


def test_wgrad_strided_operands():
    """Operands that are column slices of wider buffers (row stride != width), as the packed QKV gradient is."""
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    big_dy = torch.randn(3000, 3 * 256, generator=g).bfloat16().to(dev)
    big_x = torch.randn(3000, 512 + 64, generator=g).bfloat16().to(dev)
    dy, x = big_dy[:, 256:512], big_x[:, 64:]
    ref = dy.float().t() @ x.float()
    dw = Kn.wgrad(dy, x)
    assert (dw - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


def test_wgrad_linear_function_matches_f_linear():
    """The autograd wrapper used for the attention output projections (bias-free Linear, HIP weight gradient)."""
    import torch.nn.functional as F

    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    lin = torch.nn.Linear(256, 128).to(dev)
    x0 = torch.randn(40, 500, 256, device=dev)
    w = torch.randn(40, 500, 128, device=dev)
    outs = []
    for custom in (False, True):
        lin.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert fused._wgrad_linear_ok(lin.weight, x)
            y = fused.linear(x, lin.weight, lin.bias) if custom else F.linear(x, lin.weight, lin.bias)
        assert y.dtype == torch.bfloat16
        (y.float() * w).sum().backward()
        outs.append((y.float().detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
    for a, b, name in zip(outs[0], outs[1], ("y", "dx", "dw", "db")):
        assert a.dtype == b.dtype
        assert (a - b).abs().max() <= 2e-2 * a.abs().max(), name


@pytest.mark.parametrize("n,k,dtype", [(768, 3072, torch.float32), (2304, 768, torch.float32), (100, 36, torch.float32),
                                       (64, 64, torch.bfloat16), (132, 260, torch.float16), (8, 8, torch.float32)])
def test_cast_transpose_is_the_rounded_weight_and_its_transpose(n, k, dtype):
    """One pass over the master weight: bf16 copy (what autocast's cast produces, bit for bit) and its transpose."""
    from mmlearn_amd import kernels as K

    dev = torch.device("cuda", 0)
    torch.manual_seed(n + k)
    w = torch.randn(n, k, device=dev).to(dtype)
    w16, w16t = K.cast_transpose(w)
    ref = w.to(torch.bfloat16)
    assert w16.shape == (n, k) and w16t.shape == (k, n) and w16.dtype == w16t.dtype == torch.bfloat16
    assert torch.equal(w16, ref)
    assert torch.equal(w16t, ref.t().contiguous())
    none, only_t = K.cast_transpose(w, want_plain=False)
    assert none is None and torch.equal(only_t, w16t)


def test_linear_backward_on_the_transposed_twin_equals_the_plain_layout(monkeypatch):
    """dX = F.linear(dY, W16^T) (default) against dX = dY @ W16 (MMK_NO_DX_TWIN=1): same products, the library may pick
    another kernel for the other layout, so the comparison is to bf16 rounding; forward and dW are the same code."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    lin = torch.nn.Linear(768, 2304).to(dev)
    x0 = torch.randn(32, 197, 768, device=dev)
    g = torch.randn(32, 197, 2304, device=dev)
    outs = []
    for plain in (True, False):
        if plain:
            monkeypatch.setenv("MMK_NO_DX_TWIN", "1")
        else:
            monkeypatch.delenv("MMK_NO_DX_TWIN")
        lin.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fused.linear(x, lin.weight, lin.bias)
        (y.float() * g).sum().backward()
        outs.append((y.float().detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    assert (outs[0][1] - outs[1][1]).abs().max() <= 1e-2 * outs[0][1].abs().max()
