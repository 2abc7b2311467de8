"""GPU: windowed self-attention (csrc/window_attention.hip) -- the attention of HTSAT, the audio tower of BASELINE configs[3] (HF
ClapAudioSelfAttention; Swin's is the same code) -- against a float32 torch restatement of the HF forward on the same bf16
projections, and the patched HF modules against the stock ones (outputs and every parameter gradient)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(q, k, v, bias, mask, heads):
    """HF ClapAudioSelfAttention.forward after the projections, in float32: q/k/v [Bw, 64, C], bias [H, 64, 64], mask [nW, 64, 64] | None."""
    Bw, T, C = q.shape
    dh = C // heads
    qh, kh, vh = (t.view(Bw, T, heads, dh).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh) + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        s = (s.view(Bw // nW, nW, heads, T, T) + mask[None, :, None]).view(Bw, heads, T, T)
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(Bw, T, C)


@pytest.mark.parametrize("B,nW,heads,dh,masked", [(3, 4, 4, 24, True), (2, 16, 8, 24, False), (5, 1, 32, 24, False), (2, 4, 3, 32, True),
                                                  (16, 64, 4, 24, True), (7, 2, 2, 24, True)])
def test_window_attention_forward_and_backward_match_float32_torch(B, nW, heads, dh, masked):
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(B * 100 + nW + heads)
    C = heads * dh
    q, k, v = ((torch.randn(B * nW, 64, C, generator=g) * 1.5).bfloat16().to(dev).requires_grad_(True) for _ in range(3))
    bias = (torch.randn(heads, 64, 64, generator=g) * 0.5).to(dev).requires_grad_(True)
    mask = None
    if masked:   # the shifted-window mask: 0 inside a region, -100 across regions
        region = torch.randint(0, 3, (nW, 64), generator=g)
        mask = ((region[:, :, None] != region[:, None, :]).float() * -100.0).to(dev)
    wgt = torch.randn(B * nW, 64, C, generator=g).to(dev)
    out = fused.window_attention(q, k, v, bias, mask, heads, 1.0 / math.sqrt(dh))
    assert out.dtype == torch.bfloat16 and out.shape == q.shape
    (out.float() * wgt).sum().backward()
    got = [out.detach().float(), q.grad.float(), k.grad.float(), v.grad.float(), bias.grad.float()]
    q32, k32, v32 = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    b32 = bias.detach().clone().requires_grad_(True)
    ref = _reference(q32, k32, v32, b32, mask, heads)
    (ref * wgt).sum().backward()
    want = [ref.detach(), q32.grad, k32.grad, v32.grad, b32.grad]
    for name, a, b in zip(("o", "dq", "dk", "dv", "dbias"), got, want):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 1e-2 * scale, (name, (a - b).abs().max().item(), scale)   # bf16 outputs / bf16 P, dS operands


def test_window_attention_is_exact_on_a_one_hot_problem():
    """Keys one-hot in the head dim, queries that select one key each with a large logit, V holding integers: the output rows are
    exact copies of single V rows, so a wrong row / column map of the MFMA operands or of the transposed reads cannot hide."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    heads, dh, Bw = 2, 24, 2
    C = heads * dh
    # the bias does the selecting: query i attends key perm[i], everything else is at -1000
    g = torch.Generator().manual_seed(3)
    perm = torch.stack([torch.randperm(64, generator=g) for _ in range(heads)])
    bias = torch.full((heads, 64, 64), -1000.0)      # exp underflows to exactly 0 for the keys not selected
    for h in range(heads):
        bias[h, torch.arange(64), perm[h]] = 0.0
    q = torch.zeros(Bw, 64, C)
    k = torch.zeros(Bw, 64, C)
    v = (torch.arange(Bw * 64 * C).view(Bw, 64, C) % 251 - 125).float()
    out = fused.window_attention(q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev), bias.to(dev), None, heads, 1.0)
    want = torch.stack([torch.cat([v[b, perm[h], h * dh:(h + 1) * dh] for h in range(heads)], dim=-1) for b in range(Bw)])
    diff = (out.float().cpu() - want.bfloat16().float()).abs()
    assert diff.max().item() == 0.0, (int((diff > 0).sum()), diff.max().item(), torch.nonzero(diff > 0)[:8].tolist())


@pytest.mark.parametrize("shift", [0, 4, 3])
def test_token_map_mode_is_roll_partition_attention_reverse_roll(shift):
    """``grid=(h, w)``: the kernel gathers the 8 x 8 windows of the map rolled by ``-shift`` and scatters the context back.  Against the
    same kernel on windows that torch rolled and partitioned (HF ClapAudioLayer.forward's copies): the same arithmetic on the same
    numbers, so outputs and all four gradients are bit-identical."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11 + shift)
    B, gh, gw, heads, dh = 3, 16, 24, 2, 24
    C, nW = heads * dh, (gh // 8) * (gw // 8)
    maps = [torch.randn(B, gh * gw, C, generator=g).bfloat16().to(dev).requires_grad_(True) for _ in range(3)]
    bias = (torch.randn(heads, 64, 64, generator=g) * 0.5).to(dev).requires_grad_(True)
    region = torch.randint(0, 3, (nW, 64), generator=g)
    mask = ((region[:, :, None] != region[:, None, :]).float() * -100.0).to(dev) if shift else None
    wgt = torch.randn(B, gh * gw, C, generator=g).to(dev)

    def part(t):    # roll + window_partition of HF
        t = torch.roll(t.view(B, gh, gw, C), shifts=(-shift, -shift), dims=(1, 2))
        return t.view(B, gh // 8, 8, gw // 8, 8, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, 64, C)

    def unpart(t):  # window_reverse + roll back
        t = t.view(B, gh // 8, gw // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, gh, gw, C)
        return torch.roll(t, shifts=(shift, shift), dims=(1, 2)).view(B, gh * gw, C)

    res = []
    for mode in ("map", "copies"):
        for t in maps + [bias]:
            t.grad = None
        if mode == "map":
            out = fused.window_attention(*maps, bias, mask, heads, 1.0 / math.sqrt(dh), (gh, gw), shift)
        else:
            out = unpart(fused.window_attention(*(part(t) for t in maps), bias, mask, heads, 1.0 / math.sqrt(dh)))
        (out.float() * wgt).sum().backward()
        res.append([out.detach().clone()] + [t.grad.clone() for t in maps] + [bias.grad.clone()])
    for name, a, b in zip(("o", "dq", "dk", "dv"), res[0], res[1]):
        assert torch.equal(a, b), name
    assert (res[0][4] - res[1][4]).abs().max().item() <= 1e-5 * max(1.0, res[1][4].abs().max().item())   # partial sums are grouped differently


@pytest.mark.parametrize("grid,shift", [(None, 0), ((16, 16), 5)])
def test_packed_projection_mode_equals_three_separate_tensors(grid, shift):
    """q | k | v as the thirds of one ``[..., 3 C]`` tensor (row stride 3 C in the kernels, the three gradients written into one packed
    buffer): bit-identical to the same call on three contiguous tensors, in window mode and in token-map mode."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    heads, dh, B = 3, 24, 2
    C, nW = heads * dh, 4
    shape = (B * nW, 64, 3 * C) if grid is None else (B, grid[0] * grid[1], 3 * C)
    if grid is not None:
        nW = (grid[0] // 8) * (grid[1] // 8)
    qkv = torch.randn(*shape, generator=g).bfloat16().to(dev).requires_grad_(True)
    bias = (torch.randn(heads, 64, 64, generator=g) * 0.5).to(dev).requires_grad_(True)
    region = torch.randint(0, 3, (nW, 64), generator=g)
    mask = ((region[:, :, None] != region[:, None, :]).float() * -100.0).to(dev)
    wgt = torch.randn(*shape[:-1], C, generator=g).to(dev)
    out_p = fused.window_attention_packed(qkv, bias, mask, heads, 0.2, grid, shift)
    (out_p.float() * wgt).sum().backward()
    g_p, gb_p = qkv.grad.clone(), bias.grad.clone()
    qkv.grad = bias.grad = None
    q, k, v = (qkv[..., i * C:(i + 1) * C].contiguous() for i in range(3))
    out_s = fused.window_attention(q, k, v, bias, mask, heads, 0.2, grid, shift)
    (out_s.float() * wgt).sum().backward()
    assert torch.equal(out_p, out_s) and torch.equal(g_p, qkv.grad) and torch.equal(gb_p, bias.grad)


def test_patched_clap_audio_layers_match_the_stock_modules():
    """HF ClapAudioModel (HTSAT, head dim 24, shifted and unshifted layers, four resolutions) with and without the fused windowed
    attention: pooled output and every parameter gradient.  eval() so that the two passes see the same network (no dropout / drop
    path draws; gradients flow all the same); the 4 x 4 / stride 4 patch convolution runs as im2col + GEMM in both passes --
    MIOpen's bf16 kernels for that convolution fault intermittently on this stack, which a test must not provoke."""
    from transformers import ClapAudioConfig, ClapAudioModel

    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = ClapAudioConfig(depths=(2, 2, 2, 1), num_attention_heads=(2, 4, 8, 16), patch_embeds_hidden_size=48, hidden_size=384)   # head dim 24
    model = ClapAudioModel(cfg).to(dev).eval()
    assert fused.patch_conv_as_gemm(model) >= 1
    for m in model.modules():   # the bias tables are zero-initialised in HF: give them values so their gradient path is exercised
        if hasattr(m, "relative_position_bias_table"):
            torch.nn.init.normal_(m.relative_position_bias_table, std=0.5)
    x = torch.randn(3, 1, 1001, 64, device=dev)

    def run():
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(input_features=x).pooler_output
        out.float().square().sum().backward()
        return out.detach().float(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}

    o0, g0 = run()
    assert fused.fuse_window_attention(model) == 7 and fused.patch_mel_stretch(model) == 1
    o1, g1 = run()
    assert (o0 - o1).abs().max().item() <= 3e-2 * max(1.0, o0.abs().max().item())
    assert g0.keys() == g1.keys() and any("relative_position_bias_table" in name for name in g1)
    for name in g0:
        if name.endswith("self.key.bias"):
            continue   # softmax is invariant to a constant added to every key's logit: the true gradient is 0, both passes hold rounding noise
        scale = max(1e-3, g0[name].abs().max().item())
        assert (g0[name] - g1[name]).abs().max().item() <= 5e-2 * scale, (name, (g0[name] - g1[name]).abs().max().item(), scale)


@pytest.mark.parametrize("n,h_in,h_out,w", [(3, 1001, 1024, 64), (2, 10, 37, 8), (1, 2, 5, 4), (4, 300, 301, 12)])
def test_one_axis_bicubic_stretch_matches_aten(n, h_in, h_out, w):
    """``kernels.cubic_resize_rows`` against ``F.interpolate(mode="bicubic", align_corners=True)`` on an input whose last axis keeps its
    length (HTSAT's spectrogram stretch 1001 -> 1024 frames), forward and backward: the same taps in the same order, so equal to f32 rounding."""
    from mmlearn_amd import fused

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n + h_in)
    x = torch.randn(n, 1, h_in, w, generator=g).to(dev).requires_grad_(True)
    wgt = torch.randn(n, 1, h_out, w, generator=g).to(dev)
    y = fused._CubicRowsFn.apply(x, h_out)
    (y * wgt).sum().backward()
    gx = x.grad.clone()
    x.grad = None
    ref = torch.nn.functional.interpolate(x, (h_out, w), mode="bicubic", align_corners=True)
    (ref * wgt).sum().backward()
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert (gx - x.grad).abs().max().item() <= 1e-5 * max(1.0, x.grad.abs().max().item())

