"""GPU parity of the short-sequence attention kernel against a plain PyTorch f32 reference of the same op."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, scale):
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return (p @ v.float()).transpose(1, 2)  # [B, L, H, dh]


@pytest.mark.parametrize("B,H,L", [(3, 4, 197), (2, 12, 77), (2, 2, 256), (1, 3, 32), (2, 2, 169), (5, 1, 1), (2, 3, 100)])
def test_attention_fwd_bwd_vs_torch(B, H, L):
    from mmlearn_amd.attention import attention

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(B * 1000 + L)
    dh = 64
    # the HF layout: [B, L, H*dh] projections viewed as [B, H, L, dh]
    def mk():
        return (torch.randn(B, L, H * dh, generator=g) * 1.5).bfloat16()

    q0, k0, v0 = mk(), mk(), mk()
    scale = dh ** -0.5
    qr, kr, vr = (t.float().view(B, L, H, dh).transpose(1, 2).clone().requires_grad_(True) for t in (q0, k0, v0))
    out_r = _ref(qr, kr, vr, scale)
    w = torch.randn(B, L, H, dh, generator=g)
    (out_r * w).sum().backward()

    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q0, k0, v0))
    out = attention(qd.view(B, L, H, dh).transpose(1, 2), kd.view(B, L, H, dh).transpose(1, 2), vd.view(B, L, H, dh).transpose(1, 2), scale)
    assert out.shape == (B, L, H, dh) and out.is_contiguous()
    err = (out.float().cpu() - out_r.detach()).abs().max().item()
    assert err <= 2e-2 * max(1.0, out_r.abs().max().item()), err
    (out.float() * w.to(dev)).sum().backward()
    for got, ref, name in ((qd.grad, qr.grad, "dq"), (kd.grad, kr.grad, "dk"), (vd.grad, vr.grad, "dv")):
        ref_l = ref.transpose(1, 2).reshape(B, L, H * dh)
        e = (got.float().cpu() - ref_l).abs().max().item()
        assert e <= 3e-2 * max(ref_l.abs().max().item(), 1e-3), (name, e, ref_l.abs().max().item())


@pytest.mark.parametrize("B,H,L,p", [(2, 3, 77, 0.1), (2, 2, 197, 0.1), (1, 2, 64, 0.5), (3, 1, 33, 0.25)])
def test_attention_dropout_matches_oracle_mask(B, H, L, p):
    """The kernel's keep-mask is a pure function of (seed, b, h, i, j); oracle/attention_oracle.py restates it in numpy,
    so forward and backward with dropout are compared exactly like the p = 0 case."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import attention_oracle as AO
    from mmlearn_amd.attention import attention

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(17 * L + B)
    dh, scale, seed = 64, 0.125, 0x1234_5678_9ABC_DEF0 + L
    q0, k0, v0 = ((torch.randn(B, L, H * dh, generator=g) * 1.5).bfloat16() for _ in range(3))
    qr, kr, vr = (t.float().view(B, L, H, dh).transpose(1, 2).clone().requires_grad_(True) for t in (q0, k0, v0))
    out_r = AO.attention(qr, kr, vr, scale, p, seed)
    w = torch.randn(B, L, H, dh, generator=g)
    (out_r * w).sum().backward()
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q0, k0, v0))
    out = attention(*(t.view(B, L, H, dh).transpose(1, 2) for t in (qd, kd, vd)), scale, p, seed)
    err = (out.float().cpu() - out_r.detach()).abs().max().item()
    assert err <= 2e-2 * max(1.0, out_r.abs().max().item()), err
    (out.float() * w.to(dev)).sum().backward()
    for got, ref, name in ((qd.grad, qr.grad, "dq"), (kd.grad, kr.grad, "dk"), (vd.grad, vr.grad, "dv")):
        ref_l = ref.transpose(1, 2).reshape(B, L, H * dh)
        e = (got.float().cpu() - ref_l).abs().max().item()
        assert e <= 3e-2 * max(ref_l.abs().max().item(), 1e-3), (name, e, ref_l.abs().max().item())
    # the realised keep rate is what the threshold promises, and a different seed gives a different mask
    m = AO.keep_mask(seed, B, H, L, p)
    assert abs(m.mean() - (1 - AO.drop_threshold(p) / 65536)) < 4 * np.sqrt(p * (1 - p) / m.size) + 1e-3
    out2 = attention(*(t.detach().view(B, L, H, dh).transpose(1, 2) for t in (qd, kd, vd)), scale, p, seed + 1)
    assert (out2.float() - out.float()).abs().max().item() > 1e-3


def test_attention_lse_and_hf_interface():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from mmlearn_amd import kernels as K
    from mmlearn_amd.attention import register_hf_attention

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    q = torch.randn(2, 3, 197, 64, device=dev).bfloat16()
    k = torch.randn(2, 3, 197, 64, device=dev).bfloat16()
    v = torch.randn(2, 3, 197, 64, device=dev).bfloat16()
    _, lse = K.attn_fwd(q, k, v, 0.125)
    ref = torch.logsumexp((q.float() @ k.float().transpose(-1, -2)) * 0.125, dim=-1)
    np.testing.assert_allclose(lse.cpu().numpy(), ref.cpu().numpy(), atol=2e-3)

    name = register_hf_attention()
    cfg = CLIPVisionConfig(patch_size=16, image_size=224, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2, projection_dim=64)
    m_ref = CLIPVisionModelWithProjection(cfg).to(dev)
    cfg2 = CLIPVisionConfig(**{**cfg.to_dict()})
    cfg2._attn_implementation = name
    m_hip = CLIPVisionModelWithProjection(cfg2).to(dev)
    m_hip.load_state_dict(m_ref.state_dict())
    px = torch.rand(3, 3, 224, 224, device=dev)
    outs = []
    for m in (m_ref, m_hip):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            e = m(pixel_values=px).image_embeds
        e.float().square().mean().backward()
        outs.append((e.float().detach(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    (e0, g0), (e1, g1) = outs
    assert (e0 - e1).abs().max() <= 3e-2 * e0.abs().max()
    gmax = max(v.abs().max().item() for v in g0.values())
    for n in g0:
        assert (g0[n] - g1[n]).abs().max() <= 6e-2 * max(g0[n].abs().max().item(), 1e-2 * gmax), n


@pytest.mark.parametrize("B,H,L,p", [(2, 4, 197, 0.0), (3, 2, 77, 0.1), (1, 3, 50, 0.0)])
def test_attention_qkvpacked_matches_unpacked(B, H, L, p):
    """The packed entry (one [B, L, 3, H, 64] tensor in, one packed gradient out) is the same computation."""
    from mmlearn_amd.attention import attention, attention_qkvpacked

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5 * L + H)
    qkv0 = (torch.randn(B, L, 3, H, 64, generator=g) * 1.5).bfloat16().to(dev)
    w = torch.randn(B, L, H, 64, generator=g).to(dev)
    a = qkv0.clone().requires_grad_(True)
    out_a = attention_qkvpacked(a, 0.125, p, 99)
    (out_a.float() * w).sum().backward()
    b = qkv0.clone().requires_grad_(True)
    q, k, v = (b[:, :, i].transpose(1, 2) for i in range(3))
    out_b = attention(q, k, v, 0.125, p, 99)
    (out_b.float() * w).sum().backward()
    assert torch.equal(out_a, out_b)
    assert torch.equal(a.grad, b.grad)


def test_fused_qkv_modules_match_stock_hf_models():
    """fuse_qkv_attention keeps parameters / state_dict and reproduces the stock CLIP and BERT encoders (dropout off)."""
    from transformers import BertConfig, BertModel, CLIPVisionConfig, CLIPVisionModelWithProjection

    from mmlearn_amd.fused import fuse_qkv_attention

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(patch_size=16, image_size=224, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2, projection_dim=64)
    bcfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cases = [(CLIPVisionModelWithProjection(cfg).to(dev), {"pixel_values": torch.rand(3, 3, 224, 224, device=dev)}, "image_embeds"),
             (BertModel(bcfg, add_pooling_layer=False).to(dev), {"input_ids": torch.randint(0, 30522, (4, 77), device=dev)}, "last_hidden_state")]
    for model, inputs, field in cases:
        keys = list(model.state_dict().keys())
        outs = []
        for fused in (False, True):
            if fused:
                assert fuse_qkv_attention(model) == 2
                assert list(model.state_dict().keys()) == keys
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                e = getattr(model(**inputs), field)
            e.float().square().mean().backward()
            outs.append((e.float().detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        (e0, g0), (e1, g1) = outs
        assert (e0 - e1).abs().max() <= 3e-2 * e0.abs().max()
        assert g0.keys() == g1.keys()
        gmax = max(v.abs().max().item() for v in g0.values())
        for n in g0:
            assert (g0[n] - g1[n]).abs().max() <= 6e-2 * max(g0[n].abs().max().item(), 1e-2 * gmax), n


@pytest.mark.parametrize("B,H,L,p", [(3, 2, 197, 0.0), (2, 3, 77, 0.1), (2, 2, 256, 0.0), (4, 1, 1, 0.0), (2, 2, 130, 0.0), (1, 2, 33, 0.0)])
def test_packed_backward_column_sums_equal_the_sum_over_the_stored_gradient(B, H, L, p):
    """mmk_attn_bwd(colsum_part): the per-tile sums taken while dq / dk / dv are stored add up to dqkv.sum over rows --
    the fused QKV projection's bias gradient -- for every kernel variant (5- and 7-product, 16-row staging at L = 256)."""
    from mmlearn_amd import kernels as K
    g = torch.Generator().manual_seed(L * 7 + H)
    qkv = (torch.randn(B, L, 3, H, 64, generator=g) * 1.3).bfloat16().cuda()
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    out, lse = K.attn_fwd(q, k, v, 0.125, p, 99)
    dout = torch.randn(B, L, H, 64, generator=g).bfloat16().cuda()
    plain = K.attn_bwd(q, k, v, out, lse, dout, 0.125, p, 99, packed=True)
    dqkv, cs = K.attn_bwd(q, k, v, out, lse, dout, 0.125, p, 99, packed=True, colsum=True)
    assert torch.equal(plain, dqkv)                      # asking for the sums does not change the gradient
    want = dqkv.view(B * L, -1).double().sum(0)
    assert cs.shape == (3 * H * 64,) and cs.dtype == torch.float32
    scale = max(1.0, want.abs().max().item())
    assert (cs.double() - want).abs().max().item() <= 1e-5 * scale * max(1.0, (B * L) ** 0.5 / 8), (cs.double() - want).abs().max().item()


@pytest.mark.parametrize("B,H,L,p", [(6, 12, 197, 0.0), (5, 3, 77, 0.1), (2, 2, 256, 0.0), (3, 4, 1, 0.0), (4, 2, 65, 0.25), (1025, 12, 50, 0.0)])
def test_single_query_attention_matches_row_0_of_the_oracle(B, H, L, p):
    """csrc/cls_attention.hip (the token-0 query of a tower's last layer, fused.cls_only_last_layer): output, d q, and EVERY row of
    d k / d v against row 0 of the full attention oracle (float32 torch + the numpy restatement of the dropout mask at query index 0)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import attention_oracle as AO
    from mmlearn_amd.fused import _ClsAttnFn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(31 * L + B)
    dh, scale, seed = 64, 0.125, 0x0BAD_5EED_0000_0000 + L
    q0 = (torch.randn(B, H, dh, generator=g) * 1.5).bfloat16()
    kv0 = (torch.randn(B, L, 2, H, dh, generator=g) * 1.5).bfloat16()
    w = torch.randn(B, H, dh, generator=g)
    Bo = min(B, 8)   # the oracle walks (sample, head) pairs in python: the first samples are enough, the kernel's grid covers all of them
    qr = torch.zeros(Bo, H, L, dh)
    qr[:, :, 0] = q0[:Bo].float()
    qr.requires_grad_(True)
    kr = kv0[:Bo, :, 0].float().transpose(1, 2).clone().requires_grad_(True)    # [B, H, L, dh]
    vr = kv0[:Bo, :, 1].float().transpose(1, 2).clone().requires_grad_(True)
    out_r = AO.attention(qr, kr, vr, scale, p, seed)[:, 0]                      # [B, H, dh]: query 0
    (out_r * w[:Bo]).sum().backward()
    qd, kvd = q0.to(dev).requires_grad_(True), kv0.to(dev).requires_grad_(True)
    out = _ClsAttnFn.apply(qd, kvd, scale, p, seed)
    assert out.shape == (B, H, dh) and out.dtype == torch.bfloat16
    err = (out[:Bo].float().cpu() - out_r.detach()).abs().max().item()
    assert err <= 2e-2 * max(1.0, out_r.abs().max().item()), err
    (out.float() * w.to(dev)).sum().backward()
    dq_ref = qr.grad[:, :, 0]
    assert (qd.grad[:Bo].float().cpu() - dq_ref).abs().max().item() <= 3e-2 * max(dq_ref.abs().max().item(), 1e-3)
    for part, ref, name in ((0, kr.grad, "dk"), (1, vr.grad, "dv")):
        ref_l = ref.transpose(1, 2)                                            # [B, L, H, dh]
        e = (kvd.grad[:Bo, :, part].float().cpu() - ref_l).abs().max().item()
        assert e <= 3e-2 * max(ref_l.abs().max().item(), 1e-3), (name, e, ref_l.abs().max().item())
    assert torch.isfinite(kvd.grad.float()).all() and torch.isfinite(out.float()).all()   # every (sample, head) was served
    if B > Bo:   # the samples the oracle did not walk: against the library's SDPA on the device
        with torch.no_grad():
            k, v = kvd.detach().unbind(2)
            ref = torch.nn.functional.scaled_dot_product_attention(qd.detach().unsqueeze(2).float(), k.transpose(1, 2).float(), v.transpose(1, 2).float(),
                                                                   scale=scale)[:, :, 0]
        assert (out.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


def test_cls_only_path_uses_the_single_query_kernel_and_agrees_with_sdpa(monkeypatch):
    """fused._cls_query_attention: the HIP single-query node against the same function on F.scaled_dot_product_attention
    (MMK_NO_CLS_ATTN=1), forward and all gradients, on a ViT-shaped problem."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    E, heads, B, L = 768, 12, 40, 197
    ql, kl, vl = (torch.nn.Linear(E, E).to(dev) for _ in range(3))
    x0 = torch.randn(B, L, E, device=dev)
    w = torch.randn(B, 1, E, device=dev)
    outs = []
    for hip in (False, True):
        if hip:
            monkeypatch.delenv("MMK_NO_CLS_ATTN", raising=False)
        else:
            monkeypatch.setenv("MMK_NO_CLS_ATTN", "1")
        for m in (ql, kl, vl):
            m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            a = fused._cls_query_attention(x, ql, kl, vl, heads, 0.125, 0.0)
        assert a.shape == (B, 1, E)
        (a.float() * w).sum().backward()
        outs.append([a.float().detach(), x.grad.clone()] + [p.grad.clone() for m in (ql, kl, vl) for p in m.parameters()])
    top = max(t.abs().max().item() for t in outs[0][1:])
    for k, (r, h) in enumerate(zip(*outs)):
        # the key bias' gradient is zero analytically (a constant added to every key's logit cancels in the softmax): both runs hold
        # bf16 noise there, so every tensor is compared on the scale of the real gradients
        scale = r.abs().max().item() if k == 0 else max(r.abs().max().item(), 2e-2 * top)
        assert (r - h).abs().max().item() <= 3e-2 * max(scale, 1e-3), k
