"""GPU parity of the MLP GEMMs with the activation in the epilogue (csrc/mlp_gemm.hip) against plain PyTorch f32 references:
forward  act(x W1^T + b1)  (+ the bias-free pre-activation), backward  (dY W2) * act'(pre + b1)  with its column sums."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: "quick_gelu", 1: "gelu"}


def _act(z, act):
    return z * torch.sigmoid(1.702 * z) if act == 0 else F.gelu(z)


def _act_grad(z, act):
    z = z.detach().clone().requires_grad_(True)
    _act(z, act).sum().backward()
    return z.grad


def _operands(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    return a, b


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (1024, 3072, 768), (768, 512, 3072), (256 * 9, 256 * 5, 192)])
def test_plain_product_vs_torch(M, N, K):
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    a, b = _operands(M, N, K, M + N + K)
    assert Kn.mlp_gemm_supported(M, N, K, K, K, N)
    c = Kn.mlp_gemm_plain(a.to(dev), b.to(dev))
    ref = a.float() @ b.float().t()
    err = (c.float().cpu() - ref).abs().max().item()
    assert c.dtype == torch.bfloat16 and err <= 1e-2 * max(1.0, ref.abs().max().item()), err


def test_product_is_exact_on_integer_data_with_an_asymmetric_operand():
    """A = [I | 0] against an asymmetric integer B: every output element is exact in bf16, so a swapped row / column map,
    a wrong k order inside a fragment or a misplaced tile cannot hide behind a tolerance."""
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    M, N, K = 512, 512, 512
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[torch.arange(M), (3 * torch.arange(M) + 7) % K] += 2.0
    b = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5) % 31 - 15).float()
    ref = a @ b.t()
    c = Kn.mlp_gemm_plain(a.bfloat16().to(dev), b.bfloat16().to(dev))
    assert torch.equal(c.float().cpu(), ref)


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (768, 3072, 768), (512, 512, 256)])
def test_forward_with_activation(M, N, K, act):
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    x, w = _operands(M, N, K, 11 * M + N + K + act)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(5)) * 0.2
    h, pre = Kn.mlp_gemm_fwd_act(x.to(dev), w.to(dev), bias.to(dev), act)
    ref_pre = x.float() @ w.float().t()
    ref_h = _act(ref_pre + bias, act)
    assert h.dtype == pre.dtype == torch.bfloat16
    assert (pre.float().cpu() - ref_pre).abs().max().item() <= 1e-2 * max(1.0, ref_pre.abs().max().item())
    assert (h.float().cpu() - ref_h).abs().max().item() <= 1e-2 * max(1.0, ref_h.abs().max().item())
    h2, none = Kn.mlp_gemm_fwd_act(x.to(dev), w.to(dev), bias.to(dev), act, want_pre=False)
    assert none is None and torch.equal(h2, h)
    # the same epilogue as the unfused pair (library GEMM -> bias_act kernel), up to the rounding of the pre-activation
    h_unfused = Kn.bias_act_fwd(F.linear(x.to(dev), w.to(dev)), bias.to(dev), act)
    assert (h.float() - h_unfused.float()).abs().max().item() <= 2e-2 * max(1.0, ref_h.abs().max().item())


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1024, 3072, 768), (512, 768, 256), (256 * 5, 512, 768)])
def test_backward_with_activation_gradient_and_column_sums(M, N, K, act):
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    dy, wt = _operands(M, N, K, 7 * M + N + K + act)
    g = torch.Generator().manual_seed(M + act)
    pre = (torch.randn(M, N, generator=g) * 1.5).bfloat16()
    bias = torch.randn(N, generator=g) * 0.2
    dpre, db = Kn.mlp_gemm_bwd_dact(dy.to(dev), wt.to(dev), pre.to(dev), bias.to(dev), act)
    dact = dy.float() @ wt.float().t()
    ref = dact * _act_grad(pre.float() + bias, act)
    scale = max(1.0, ref.abs().max().item())
    assert dpre.dtype == torch.bfloat16 and dpre.shape == (M, N)
    assert (dpre.float().cpu() - ref).abs().max().item() <= 1e-2 * scale
    ref_db = ref.sum(0)
    assert db.dtype == torch.float32 and db.shape == (N,)
    assert (db.cpu() - ref_db).abs().max().item() <= 2e-3 * max(1.0, ref.abs().sum(0).max().item())
    # the unfused pair it replaces
    dx_u, db_u = Kn.bias_act_bwd(pre.to(dev), bias.to(dev), F.linear(dy.to(dev), wt.to(dev)), act)
    assert (dpre.float() - dx_u.float()).abs().max().item() <= 2e-2 * scale
    assert (db - db_u).abs().max().item() <= 5e-3 * max(1.0, ref.abs().sum(0).max().item())
    dpre2, none = Kn.mlp_gemm_bwd_dact(dy.to(dev), wt.to(dev), pre.to(dev), bias.to(dev), act, want_dbias=False)
    assert none is None and torch.equal(dpre2, dpre)


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1024, 3072, 768), (512, 768, 256)])
def test_forward_that_leaves_the_derivative_and_backward_that_multiplies_by_it(M, N, K, act):
    """The pair the product runs: ``mlp_gemm_fwd_act_grad`` -> (act(z), act'(z)) for z = x W^T + b; ``mlp_gemm_bwd_mul`` ->
    (dY Wt^T) * act'(z) with its column sums.  Against f32 references and against the epilogue-form kernels."""
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    x, w = _operands(M, N, K, 3 * M + N + K + act)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(6)) * 0.2
    h, g = Kn.mlp_gemm_fwd_act_grad(x.to(dev), w.to(dev), bias.to(dev), act)
    z = x.float() @ w.float().t() + bias
    assert (h.float().cpu() - _act(z, act)).abs().max().item() <= 1e-2 * max(1.0, _act(z, act).abs().max().item())
    assert (g.float().cpu() - _act_grad(z, act)).abs().max().item() <= 8e-3          # |act'| <= 1.13, stored in bf16
    h_e, _ = Kn.mlp_gemm_fwd_act(x.to(dev), w.to(dev), bias.to(dev), act)
    assert torch.equal(h, h_e)
    # backward on another product of the same [M, N] shape
    dy, wt = _operands(M, N, 128 if K < 128 else K, 9 * M + N + act)
    dpre, db = Kn.mlp_gemm_bwd_mul(dy.to(dev), wt.to(dev), g)
    ref = (dy.float() @ wt.float().t()) * g.float().cpu()
    scale = max(1.0, ref.abs().max().item())
    assert dpre.dtype == torch.bfloat16 and (dpre.float().cpu() - ref).abs().max().item() <= 1e-2 * scale
    assert (db.cpu() - ref.sum(0)).abs().max().item() <= 3e-3 * max(1.0, ref.abs().sum(0).max().item())
    dpre2, none = Kn.mlp_gemm_bwd_mul(dy.to(dev), wt.to(dev), g, want_dbias=False)
    assert none is None and torch.equal(dpre2, dpre)


def test_strided_operands_and_many_tiles_per_workgroup():
    """Operands that are column slices of wider buffers, and more tiles than workgroups (every workgroup walks several tiles,
    the ring keeps streaming across tile boundaries)."""
    from mmlearn_amd import kernels as Kn

    dev = torch.device("cuda", 0)
    M, N, K = 256 * 24, 256 * 12, 128
    g = torch.Generator().manual_seed(9)
    big_a = torch.randn(M, K + 64, generator=g).bfloat16().to(dev)
    big_b = (torch.randn(N, K + 8, generator=g) / K ** 0.5).bfloat16().to(dev)
    a, b = big_a[:, 64:], big_b[:, :K]
    assert Kn.mlp_gemm_supported(M, N, K, a.stride(0), b.stride(0), N)
    c = Kn.mlp_gemm_plain(a, b)
    ref = a.float() @ b.float().t()
    assert (c.float() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


def test_unsupported_shapes_are_refused():
    from mmlearn_amd import kernels as Kn

    assert not Kn.mlp_gemm_supported(250, 256, 128, 128, 128, 256)
    assert not Kn.mlp_gemm_supported(256, 200, 128, 128, 128, 200)
    assert not Kn.mlp_gemm_supported(256, 256, 96, 96, 96, 256)
    dev = torch.device("cuda", 0)
    with pytest.raises(RuntimeError):
        Kn.mlp_gemm_plain(torch.zeros(250, 128, dtype=torch.bfloat16, device=dev), torch.zeros(256, 128, dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
@pytest.mark.parametrize("E,H", [(256, 512), (768, 1024)])
def test_mlp_node_matches_the_unfused_ops_and_f32_autograd(E, H, act, monkeypatch):
    """``fused.mlp_fc1_act_fc2`` (bias-free fc1, then activation + fc2 as one autograd node whose backward is the fused dX GEMM)
    against (a) the same node on library GEMM + ``bias_act_bwd`` and (b) plain f32 autograd of the same MLP."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    rows = 8192
    fc1, fc2 = torch.nn.Linear(E, H).to(dev), torch.nn.Linear(H, E).to(dev)
    x0 = torch.randn(32, rows // 32, E, device=dev)
    wgt = torch.randn(32, rows // 32, E, device=dev)

    def run(mode):
        for p in list(fc1.parameters()) + list(fc2.parameters()):
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if mode == "f32":
            z = F.linear(x, fc1.weight, fc1.bias)
            y = F.linear(_act(z, 0 if act == "quick_gelu" else 1), fc2.weight)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = fused.mlp_fc1_act_fc2(x, fc1, act, fc2)
        (y.float() * wgt).sum().backward()
        return [y.detach().float(), x.grad.float(), fc1.weight.grad.float(), fc1.bias.grad.float(), fc2.weight.grad.float()]

    fused_out = run("fused")
    monkeypatch.setenv("MMK_NO_MLP_FUSION", "1")
    unfused_out = run("unfused")
    monkeypatch.delenv("MMK_NO_MLP_FUSION")
    ref = run("f32")
    for name, a, b, c in zip(("y", "dx", "dW1", "db1", "dW2"), fused_out, unfused_out, ref):
        scale = max(1.0, c.abs().max().item())
        assert (a - c).abs().max().item() <= 3e-2 * scale, (name, "vs f32", (a - c).abs().max().item(), scale)
        assert (a - b).abs().max().item() <= 2e-2 * scale, (name, "vs unfused", (a - b).abs().max().item(), scale)
    # without gradients (evaluation) the separate ops run and give the same activations
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y_eval = fused.mlp_fc1_act_fc2(x0, fc1, act, fc2)
    assert (y_eval.float() - ref[0]).abs().max().item() <= 3e-2 * max(1.0, ref[0].abs().max().item())


@pytest.mark.parametrize("rows", [8192 + 100, 6144 + 1, 1000])
def test_mlp_node_pads_ragged_row_counts_and_the_no_grad_forward_is_fused_too(rows, monkeypatch):
    """Row counts that are not a multiple of 256 (I-JEPA: batch x kept patches): the node pads x / dY with zero rows for the fused
    GEMMs (the node itself starts at 6,144 rows, like the weight-gradient path; 1000 rows stay on the separate ops) -- output and every gradient against the unfused
    node; under ``no_grad`` (EMA teacher, evaluation) fc1 + bias + activation run as one kernel and give the same activations."""
    from mmlearn_amd import fused, kernels as Kn

    dev = torch.device("cuda", 0)
    torch.manual_seed(rows)
    E, H = 256, 512
    fc1, fc2 = torch.nn.Linear(E, H).to(dev), torch.nn.Linear(H, E).to(dev)
    x0 = torch.randn(rows, E, device=dev)
    wgt = torch.randn(rows, E, device=dev)
    calls = []
    real = Kn.mlp_gemm_fwd_act_grad
    monkeypatch.setattr(Kn, "mlp_gemm_fwd_act_grad", lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1])

    def run():
        for p in list(fc1.parameters()) + list(fc2.parameters()):
            p.grad = None
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fused.mlp_fc1_act_fc2(x, fc1, "gelu", fc2)
        (y.float() * wgt).sum().backward()
        return [y.detach().float(), x.grad.float(), fc1.weight.grad.float(), fc1.bias.grad.float(), fc2.weight.grad.float()]

    got = run()
    assert calls == ([(rows + 255) // 256 * 256] if rows >= 6144 else [])
    monkeypatch.setenv("MMK_NO_MLP_FUSION", "1")
    want = run()
    monkeypatch.delenv("MMK_NO_MLP_FUSION")
    for name, a, b in zip(("y", "dx", "dW1", "db1", "dW2"), got, want):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item()), name
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y_ng = fused.mlp_fc1_act_fc2(x0, fc1, "gelu", fc2)
    assert y_ng.shape == got[0].shape and (y_ng.float() - got[0]).abs().max().item() <= 2e-2 * max(1.0, got[0].abs().max().item())

