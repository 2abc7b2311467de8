"""Randomised shape sweep of the attention, weight-gradient, row, recall, windowed-attention and embedding-backward kernels against torch references (run by hand on the GPU)."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
from mmlearn_amd.attention import attention_qkvpacked

dev = torch.device("cuda", 0)
rng = random.Random(int(os.environ.get("SEED", 0)))
bad = 0
for it in range(int(os.environ.get("N", 60))):
    B, H, L = rng.randint(1, 6), rng.randint(1, 5), rng.randint(1, 256)
    p = rng.choice([0.0, 0.0, 0.1])
    qkv = (torch.randn(B, L, 3, H, 64, device=dev) * 1.2).bfloat16().requires_grad_(True)
    out = attention_qkvpacked(qkv, 0.125, p, 1234 + it)
    w = torch.randn_like(out, dtype=torch.float32)
    (out.float() * w).sum().backward()
    if p == 0.0:
        q, k, v = (qkv.detach()[:, :, i].transpose(1, 2).float().requires_grad_(True) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2)
        (ref * w).sum().backward()
        gref = torch.stack([t.grad.transpose(1, 2) for t in (q, k, v)], 2)
        e1 = (out.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        e2 = (qkv.grad.float() - gref).abs().max().item() / max(1e-3, gref.abs().max().item())
        if e1 > 2e-2 or e2 > 4e-2 or not torch.isfinite(qkv.grad.float()).all():
            bad += 1; print("ATTN MISMATCH", B, H, L, e1, e2)
    elif not (torch.isfinite(out.float()).all() and torch.isfinite(qkv.grad.float()).all()):
        bad += 1; print("ATTN NONFINITE", B, H, L, p)
for it in range(int(os.environ.get("N", 60))):
    M, N, Kk = rng.randint(1, 9000), 8 * rng.randint(1, 120), 8 * rng.randint(1, 120)
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Kk, device=dev).bfloat16()
    ref = dy.float().t() @ x.float()
    got = K.wgrad(dy, x)
    e = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    if e > 3e-3:
        bad += 1; print("WGRAD MISMATCH", M, N, Kk, e)
torch.cuda.synchronize()
print("fuzz done, mismatches:", bad)

# ---- row kernels: add + LayerNorm (+ deferred bias, dropout off) and bias + activation
from mmlearn_amd import fused
import torch.nn.functional as F
bad2 = 0
for it in range(int(os.environ.get("N", 60)) // 2):
    rows, d = rng.randint(1, 700), 8 * rng.randint(1, 256)
    ln = fused.LayerNorm.from_torch(torch.nn.LayerNorm(d).to(dev), False)
    x = torch.randn(rows, d, device=dev).bfloat16().requires_grad_(True)
    r = torch.randn(rows, d, device=dev).requires_grad_(True)
    xb = torch.randn(d, device=dev).requires_grad_(True)
    s, y = fused.add_layer_norm(x, r, ln, xbias=xb)
    ws, wy = torch.randn_like(s), torch.randn_like(y)
    ((s * ws).sum() + (y * wy).sum()).backward()
    x2, r2, xb2 = x.detach().float().requires_grad_(True), r.detach().clone().requires_grad_(True), xb.detach().clone().requires_grad_(True)
    s_ref = r2 + x2 + xb2
    y_ref = F.layer_norm(s_ref, (d,), ln.weight.detach(), ln.bias.detach(), ln.eps)
    ((s_ref * ws).sum() + (y_ref * wy).sum()).backward()
    errs = [(s - s_ref).abs().max().item(), (y - y_ref).abs().max().item(), (x.grad.float() - x2.grad).abs().max().item() / max(1.0, x2.grad.abs().max().item()),
            (r.grad - r2.grad).abs().max().item() / max(1.0, r2.grad.abs().max().item()), (xb.grad - xb2.grad).abs().max().item() / max(1.0, xb2.grad.abs().max().item())]
    if max(errs[:2]) > 1e-3 or max(errs[2:]) > 2e-2:
        bad2 += 1; print("ADDLN MISMATCH", rows, d, errs)
    act = rng.choice(["quick_gelu", "gelu"])
    z = torch.randn(rows, d, device=dev).bfloat16().requires_grad_(True)
    bb = torch.randn(d, device=dev).requires_grad_(True)
    a = fused.bias_act(z, bb, act)
    wa = torch.randn_like(a, dtype=torch.float32)
    (a.float() * wa).sum().backward()
    z2, b2 = z.detach().float().requires_grad_(True), bb.detach().clone().requires_grad_(True)
    t = z2 + b2
    a_ref = t * torch.sigmoid(1.702 * t) if act == "quick_gelu" else F.gelu(t)
    (a_ref * wa).sum().backward()
    e = [(a.float() - a_ref).abs().max().item() / max(1.0, a_ref.abs().max().item()), (z.grad.float() - z2.grad).abs().max().item() / max(1.0, z2.grad.abs().max().item()),
         (bb.grad - b2.grad).abs().max().item() / max(1.0, b2.grad.abs().max().item())]
    if max(e[:2]) > 2e-2 or e[2] > 5e-2:
        bad2 += 1; print("BIASACT MISMATCH", rows, d, act, e)
torch.cuda.synchronize()
print("row-kernel fuzz done, mismatches:", bad2)

# ---- rank-of-positive kernel (retrieval recall / zero-shot top-k): exact integer ranks vs a float64 count, ties by index
bad3 = 0
for it in range(int(os.environ.get("N", 60)) // 2):
    n, m, d = rng.randint(1, 700), rng.randint(1, 900), rng.randint(1, 300)
    x = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1)
    y = torch.nn.functional.normalize(torch.randn(m, d, device=dev), dim=-1)
    if m > 4 and rng.random() < 0.5:   # exact duplicates in the database: the tie rule decides
        y[rng.randrange(m)] = y[rng.randrange(m)]
    pos = torch.randint(0, m, (n,), device=dev)
    got = K.recall_ranks(x, y, pos)
    s = x.double() @ y.double().T
    t = s.gather(1, pos[:, None])
    col = torch.arange(m, device=dev)[None, :]
    margin = (s - t).abs()
    margin[torch.arange(n, device=dev), pos] = 1.0
    safe = margin.min(dim=1).values > 1e-5    # f32 vs f64 rounding can flip a comparison only on near-ties (duplicates excepted)
    dup = (s == t) & (col != pos[:, None])
    want = ((s > t) | ((s == t) & (col < pos[:, None]))).sum(1)
    ok = (got.long() == want) | ~(safe | dup.any(1))
    if not bool(ok.all()):
        bad3 += 1; print("RECALL MISMATCH", n, m, d, int((~ok).sum()))
torch.cuda.synchronize()
print("fuzz recall kernel done, mismatches:", bad3)

# ---- windowed attention (window mode and token-map mode) and the sorted-run embedding backward
import math
bad4 = 0
for it in range(int(os.environ.get("N", 60)) // 2):
    heads, dh = rng.randint(1, 12), rng.choice([24, 24, 32])
    gh, gw = 8 * rng.randint(1, 4), 8 * rng.randint(1, 4)
    nW, B, C = (gh // 8) * (gw // 8), rng.randint(1, 9), heads * dh
    shift = rng.choice([0, 0, rng.randint(1, 7)])
    use_map = rng.random() < 0.5
    q, k, v = ((torch.randn(B, gh * gw, C, device=dev) * 1.3).bfloat16().requires_grad_(True) for _ in range(3))
    bias = (torch.randn(heads, 64, 64, device=dev) * 0.5).requires_grad_(True)
    region = torch.randint(0, 3, (nW, 64), device=dev)
    mask = (region[:, :, None] != region[:, None, :]).float() * -100.0 if (shift or rng.random() < 0.3) else None
    w = torch.randn(B, gh * gw, C, device=dev)
    part = lambda t: torch.roll(t.view(B, gh, gw, C), (-shift, -shift), (1, 2)).view(B, gh // 8, 8, gw // 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, C)
    unpart = lambda t: torch.roll(t.view(B, gh // 8, gw // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, gh, gw, C), (shift, shift), (1, 2)).view(B, gh * gw, C)
    if use_map:
        out = fused.window_attention(q, k, v, bias, mask, heads, 1 / math.sqrt(dh), (gh, gw), shift)
    else:
        out = unpart(fused.window_attention(part(q), part(k), part(v), bias, mask, heads, 1 / math.sqrt(dh)))
    (out.float() * w).sum().backward()
    q2, k2, v2, b2 = (t.detach().float().requires_grad_(True) for t in (q, k, v, bias))
    qh, kh, vh = (part(t).view(-1, 64, heads, dh).transpose(1, 2) for t in (q2, k2, v2))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh) + b2[None]
    if mask is not None:
        s = (s.view(B, nW, heads, 64, 64) + mask[None, :, None]).view(-1, heads, 64, 64)
    ref = unpart((torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(-1, 64, C))
    (ref * w).sum().backward()
    errs = [(a.float() - b).abs().max().item() / max(1.0, b.abs().max().item())
            for a, b in ((out, ref), (q.grad, q2.grad), (k.grad, k2.grad), (v.grad, v2.grad), (bias.grad, b2.grad))]
    if max(errs) > 2e-2 or not all(math.isfinite(e) for e in errs):
        bad4 += 1; print("WINDOW-ATTN MISMATCH", B, gh, gw, heads, dh, shift, use_map, mask is not None, errs)
for it in range(int(os.environ.get("N", 60)) // 3):
    rows, d, vocab = rng.randint(4096, 120000), 4 * rng.randint(1, 300), rng.choice([2, 7, 1000, 30522])
    kind = rng.choice(["uniform", "hot", "runs"])
    ids = torch.randint(0, vocab, (rows,), device=dev)
    if kind == "hot":
        ids[torch.rand(rows, device=dev) < 0.7] = rng.randrange(vocab)
    elif kind == "runs":
        ids = ids[:: rng.randint(2, 50)].repeat_interleave(rng.randint(2, 50))[:rows]; rows = ids.numel()
    ids[torch.rand(rows, device=dev) < 0.05] = -1
    dout = torch.randn(rows, d, device=dev)
    keep = ids >= 0
    ref = torch.zeros(vocab, d, device=dev, dtype=torch.float64).index_add_(0, ids[keep], dout[keep].double())
    got = K.embedding_bwd(dout, ids, vocab)
    e = (got.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    if e > 1e-5:
        bad4 += 1; print("EMBEDDING-BWD MISMATCH", rows, d, vocab, kind, e)
torch.cuda.synchronize()
print("fuzz window attention / embedding backward done, mismatches:", bad4)
