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


# ------------------------------------------------------------------------------------------------ key-padding masks / causal
def _oracle():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import attention_oracle as AO
    return AO


def _lengths(B, L, g):
    """Per-sample valid lengths: random in [1, L], with the edge cases pinned -- one key only, the full row, a tile boundary."""
    n = torch.randint(1, L + 1, (B,), generator=g)
    n[0] = 1
    if B > 1:
        n[1] = L
    if B > 2:
        n[2] = min(L, 32)
    if B > 3:
        n[3] = min(L, 33)
    return n


@pytest.mark.parametrize("B,H,L,p,causal", [(6, 3, 77, 0.0, False), (5, 2, 197, 0.0, False), (6, 2, 77, 0.1, False), (4, 2, 256, 0.0, False),
                                             (4, 2, 128, 0.0, False), (4, 1, 40, 0.25, False), (5, 2, 169, 0.1, False), (3, 2, 1, 0.0, False),
                                             (5, 2, 77, 0.0, True), (4, 2, 197, 0.1, True), (4, 2, 256, 0.0, True), (4, 2, 100, 0.0, True)])
def test_masked_attention_fwd_bwd_vs_oracle(B, H, L, p, causal):
    """a key-padding mask (random valid lengths incl. 1 and L; a second case with holes in the middle = left padding /
    arbitrary 0-1 masks) and the causal triangle inside the HIP kernels: forward, dq, dk, dv against oracle/attention_oracle.py, for the
    five-product backward (L = 77, 197, 40, 169, 100, 1), the seven-product one (L = 128, 256) and with dropout.  Gradients of masked
    keys are exactly zero."""
    AO = _oracle()
    from mmlearn_amd.attention import attention, key_bias_of

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1000 * L + 10 * B + int(causal))
    dh, scale, seed = 64, 0.125, 0x5EED_0000_1234_0000 + L
    for form in ("lengths", "holes"):
        if form == "lengths":
            n = _lengths(B, L, g)
            keep = torch.arange(L)[None, :] < n[:, None]
        else:
            keep = torch.rand(B, L, generator=g) < 0.6
            keep[:, L // 2] = True                     # at least one key per sample
            if causal:
                keep[:, 0] = True                      # ... that every query may look at
        q0, k0, v0 = ((torch.randn(B, L, H * dh, generator=g) * 1.5).bfloat16() for _ in range(3))
        qr, kr, vr = (t.float().view(B, L, H, dh).transpose(1, 2).clone().requires_grad_(True) for t in (q0, k0, v0))
        out_r = AO.attention(qr, kr, vr, scale, p, seed, key_mask=keep, causal=causal)
        w = torch.randn(B, L, H, dh, generator=g)
        (out_r * w).sum().backward()
        qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q0, k0, v0))
        kb = key_bias_of(keep.to(dev), B, L)
        assert kb.shape == (B, 256) and (kb[:, :L] == 0).cpu().eq(keep).all() and torch.isinf(kb[:, L:]).all()
        out = attention(*(t.view(B, L, H, dh).transpose(1, 2) for t in (qd, kd, vd)), scale, p, seed, key_bias=kb, causal=causal)
        err = (out.float().cpu() - out_r.detach()).abs().max().item()
        assert err <= 2e-2 * max(1.0, out_r.abs().max().item()), (form, err)
        (out.float() * w.to(dev)).sum().backward()
        for got, ref, name in ((qd.grad, qr.grad, "dq"), (kd.grad, kr.grad, "dk"), (vd.grad, vr.grad, "dv")):
            ref_l = ref.transpose(1, 2).reshape(B, L, H * dh)
            e = (got.float().cpu() - ref_l).abs().max().item()
            assert e <= 3e-2 * max(ref_l.abs().max().item(), 1e-3), (form, name, e, ref_l.abs().max().item())
        dead = ~keep[:, :, None].expand(B, L, H * dh)
        assert (kd.grad.cpu()[dead] == 0).all() and (vd.grad.cpu()[dead] == 0).all()      # a masked key receives nothing


def test_key_bias_records_from_every_mask_form():
    """mmk_attn_key_bias: lengths, bool / uint8 / int32 / int64 / float keep-masks, float32 / bf16 additive masks, strided rows and the
    HF 4-D forms attention.key_mask_view can prove; a materialised [B, 1, L, L] tensor is refused."""
    from mmlearn_amd import kernels as K
    from mmlearn_amd.attention import key_bias_of, key_mask_view

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    B, L = 7, 77
    keep = torch.rand(B, L, generator=g) < 0.7
    n = keep.sum(1).int()
    right = torch.arange(L)[None, :] < n[:, None]
    want = torch.full((B, 256), float("-inf"))
    want[:, :L] = torch.where(keep, 0.0, -1e30)
    want_r = torch.full((B, 256), float("-inf"))
    want_r[:, :L] = torch.where(right, 0.0, -1e30)
    assert torch.equal(K.attn_key_bias(n.to(dev), L=L).cpu(), want_r)
    for dt in (torch.bool, torch.uint8, torch.int32, torch.int64, torch.float32):
        assert torch.equal(K.attn_key_bias(keep.to(dt).to(dev)).cpu(), want), dt
    wide = torch.zeros(B, L + 19, dtype=torch.int64)
    wide[:, :L] = keep
    assert torch.equal(K.attn_key_bias(wide.to(dev)[:, :L]).cpu(), want)               # strided rows
    for dt in (torch.float32, torch.bfloat16):
        add = torch.zeros(B, L, dtype=dt).masked_fill(~keep, torch.finfo(dt).min)
        assert torch.equal(K.attn_key_bias(add.to(dev), additive=True).cpu(), want), dt
        assert torch.equal(key_bias_of(add.to(dev)[:, None, None, :], B, L).cpu(), want)   # HF's extended additive mask
    soft = torch.randn(B, L, generator=g)                                                 # any finite additive bias: base-2 units
    got = K.attn_key_bias(soft.to(dev), additive=True).cpu()
    assert torch.allclose(got[:, :L], soft * 1.4426950408889634, rtol=1e-6)
    m4 = keep.to(dev)[:, None, None, :]
    assert torch.equal(key_bias_of(m4, B, L).cpu(), want)
    assert torch.equal(key_bias_of(m4.expand(B, 1, L, L), B, L).cpu(), want)              # stride-0 query dim: provably a key mask
    assert torch.equal(key_bias_of(m4.expand(B, 3, L, L), B, L).cpu(), want)
    assert key_mask_view(m4.expand(B, 1, L, L).contiguous(), B, L) is None                # materialised: could hold anything
    assert key_bias_of(m4.expand(B, 1, L, L).contiguous(), B, L) is None
    assert key_mask_view(keep.to(dev)[:, None, :], B, L) is None                          # 3-D masks are per-query masks


def test_all_masked_sample_is_finite_and_does_not_poison_the_batch():
    """A sample with no valid key (an empty caption): HF's additive finfo.min convention averages V uniformly; here the forward does the
    same and the backward stays finite, so the weight gradients summed over the batch keep the other samples' values."""
    AO = _oracle()
    from mmlearn_amd.attention import attention, key_bias_of

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(9)
    B, H, L, dh = 3, 2, 77, 64
    keep = torch.ones(B, L, dtype=torch.bool)
    keep[1] = False
    keep[2, 40:] = False
    q0, k0, v0 = ((torch.randn(B, L, H * dh, generator=g)).bfloat16() for _ in range(3))
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q0, k0, v0))
    out = attention(*(t.view(B, L, H, dh).transpose(1, 2) for t in (qd, kd, vd)), 0.125, key_bias=key_bias_of(keep.to(dev), B, L))
    ref = AO.attention(*(t.float().view(B, L, H, dh).transpose(1, 2) for t in (q0, k0, v0)), 0.125, key_mask=keep)
    assert (out.float().cpu() - ref).abs().max().item() <= 2e-2                          # incl. the uniform average of sample 1
    out.float().square().sum().backward()
    for t in (qd, kd, vd):
        assert torch.isfinite(t.grad.float()).all()


@pytest.mark.parametrize("B,H,L,p", [(6, 12, 77, 0.0), (5, 3, 77, 0.1), (4, 2, 197, 0.0)])
def test_single_query_attention_with_a_key_mask(B, H, L, p):
    """csrc/cls_attention.hip with key bias records: row 0 of the masked oracle, every dk / dv row (masked rows exactly zero)."""
    AO = _oracle()
    from mmlearn_amd.attention import key_bias_of
    from mmlearn_amd.fused import _ClsAttnFn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(77 * L + B)
    dh, scale, seed = 64, 0.125, 0x0BAD_5EED_0000_1111 + L
    n = _lengths(B, L, g)
    keep = torch.arange(L)[None, :] < n[:, None]
    q0 = (torch.randn(B, H, dh, generator=g) * 1.5).bfloat16()
    kv0 = (torch.randn(B, L, 2, H, dh, generator=g) * 1.5).bfloat16()
    w = torch.randn(B, H, dh, generator=g)
    qr = torch.zeros(B, H, L, dh)
    qr[:, :, 0] = q0.float()
    qr.requires_grad_(True)
    kr = kv0[:, :, 0].float().transpose(1, 2).clone().requires_grad_(True)
    vr = kv0[:, :, 1].float().transpose(1, 2).clone().requires_grad_(True)
    out_r = AO.attention(qr, kr, vr, scale, p, seed, key_mask=keep)[:, 0]
    (out_r * w).sum().backward()
    qd, kvd = q0.to(dev).requires_grad_(True), kv0.to(dev).requires_grad_(True)
    out = _ClsAttnFn.apply(qd, kvd, scale, p, seed, key_bias_of(keep.to(dev), B, L))
    assert (out.float().cpu() - out_r.detach()).abs().max().item() <= 2e-2 * max(1.0, out_r.abs().max().item())
    (out.float() * w.to(dev)).sum().backward()
    dq_ref = qr.grad[:, :, 0]
    assert (qd.grad.float().cpu() - dq_ref).abs().max().item() <= 3e-2 * max(dq_ref.abs().max().item(), 1e-3)
    for part, ref, name in ((0, kr.grad, "dk"), (1, vr.grad, "dv")):
        ref_l = ref.transpose(1, 2)
        got = kvd.grad[:, :, part].float().cpu()
        assert (got - ref_l).abs().max().item() <= 3e-2 * max(ref_l.abs().max().item(), 1e-3), name
        assert (got[~keep] == 0).all(), name


def _text_batch(B, L, dev, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 1000, (B, L), generator=g)
    n = _lengths(B, L, g)
    n[0] = 5          # (every caption keeps its first tokens)
    mask = (torch.arange(L)[None, :] < n[:, None]).long()
    return ids.to(dev), mask.to(dev)


def test_bert_with_the_tokenizers_mask_runs_the_masked_kernels_and_matches_stock(monkeypatch):
    """A padded text batch through a fused BERT: the model-level scope takes the 2-D mask, the attention modules run the MASKED HIP
    kernels (no stock forward, i.e. no SDPA, is entered), outputs at the valid positions and all parameter gradients match the stock
    model given the same mask; with gradient checkpointing on, the mask stays HF's and the result is still the stock one."""
    from transformers import BertConfig, BertModel

    from mmlearn_amd import fused
    from mmlearn_amd import kernels as K

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=1000,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    stock = BertModel(cfg, add_pooling_layer=False).to(dev)
    import copy
    hip = copy.deepcopy(stock)
    assert fused.fuse_qkv_attention(hip) == 2 and getattr(hip, "_mmk_mask_scope", None) is not None
    ids, mask = _text_batch(6, 77, dev, 5)
    w = torch.randn(6, 77, 128, device=dev) * mask[:, :, None]
    calls = {"masked": 0, "stock": 0}
    real_fwd = K.attn_fwd

    def spy(q, k, v, scale, dropout_p=0.0, seed=0, key_bias=None, causal=False):
        calls["masked"] += key_bias is not None
        return real_fwd(q, k, v, scale, dropout_p, seed, key_bias, causal)

    monkeypatch.setattr(K, "attn_fwd", spy)
    for sa in (l.attention.self for l in hip.encoder.layer):
        orig = sa._mmk_stock_forward
        sa._mmk_stock_forward = (lambda *a, _o=orig, **k: (calls.__setitem__("stock", calls["stock"] + 1), _o(*a, **k))[1])
    outs = []
    for m in (stock, hip):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = m(input_ids=ids, attention_mask=mask).last_hidden_state
        (h.float() * w).sum().backward()
        outs.append((h.float().detach() * mask[:, :, None], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert calls == {"masked": 2, "stock": 0}, calls
    assert hip._mmk_mask_scope.key_bias is None                                  # cleared after the forward
    (e0, g0), (e1, g1) = outs
    assert (e0 - e1).abs().max() <= 3e-2 * e0.abs().max()
    gmax = max(v.abs().max().item() for v in g0.values())
    assert g0.keys() == g1.keys()
    for n in g0:
        assert (g0[n] - g1[n]).abs().max() <= 6e-2 * max(g0[n].abs().max().item(), 1e-2 * gmax), n
    # an all-ones mask and no mask: the unmasked kernels or the masked ones, never the stock forward
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = hip(input_ids=ids, attention_mask=torch.ones_like(mask)).last_hidden_state
        b = hip(input_ids=ids).last_hidden_state
    assert calls["stock"] == 0 and (a.float() - b.float()).abs().max() <= 2e-2 * b.float().abs().max()
    # gradient checkpointing: layers re-run in the backward, the mask is left to HF and the stock forward serves it
    hip.gradient_checkpointing_enable()
    hip.train()
    hip.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = hip(input_ids=ids, attention_mask=mask).last_hidden_state
    (h.float() * w).sum().backward()
    assert (h.float().detach() * mask[:, :, None] - e0).abs().max() <= 3e-2 * e0.abs().max()
    for n in g0:
        assert (g0[n] - dict(hip.named_parameters())[n].grad).abs().max() <= 6e-2 * max(g0[n].abs().max().item(), 1e-2 * gmax), n


def test_clip_text_tower_causal_plus_padding_matches_stock():
    """HF CLIP's text tower is causal (clip.py:329-346 forwards the padding mask on top): the fused modules serve ``is_causal`` with the
    scope's key bias on the masked kernels; text embeddings and parameter gradients match the stock model."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection

    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    cfg = CLIPTextConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=1000,
                         max_position_embeddings=77, projection_dim=64, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    stock = CLIPTextModelWithProjection(cfg).to(dev)
    import copy
    hip = copy.deepcopy(stock)
    assert fused.fuse_qkv_attention(hip) == 2
    ids, mask = _text_batch(6, 77, dev, 8)
    ids = ids.clamp(min=3)
    n = mask.sum(1)
    ids[torch.arange(6, device=dev), n - 1] = 2           # the EOS token the pooling looks for, at the last valid position
    ids = ids * mask
    served = []
    for sa in (l.self_attn for l in hip.text_model.encoder.layers):
        orig = sa._mmk_stock_forward
        sa._mmk_stock_forward = (lambda *a, _o=orig, **k: (served.append("stock"), _o(*a, **k))[1])
    outs = []
    for m in (stock, hip):
        for am in (mask, None):
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                e = m(input_ids=ids, attention_mask=am).text_embeds
            e.float().square().sum().backward()
            outs.append((e.float().detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert served == [], served
    for (e0, g0), (e1, g1) in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert (e0 - e1).abs().max() <= 3e-2 * e0.abs().max()
        gmax = max(v.abs().max().item() for v in g0.values())
        for k in g0:
            assert (g0[k] - g1[k]).abs().max() <= 6e-2 * max(g0[k].abs().max().item(), 1e-2 * gmax), k
