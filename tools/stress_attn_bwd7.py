"""Repeated random (B, H, L) cases for the sequence lengths served by the seven-product attention backward (97..128 and
225..256 rows): packed forward + backward vs an f32 torch reference.  Run by hand on the GPU."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd.attention import attention_qkvpacked
dev = torch.device("cuda", 0); rng = random.Random(3); bad = 0
for it in range(300):
    L = rng.choice([rng.randint(225, 256), rng.randint(97, 128), 256, 128])
    B, H = rng.randint(1, 40), rng.randint(1, 12)
    qkv = (torch.randn(B, L, 3, H, 64, device=dev) * 1.2).bfloat16().requires_grad_(True)
    out = attention_qkvpacked(qkv, 0.125, 0.0, 0)
    w = torch.randn_like(out, dtype=torch.float32)
    (out.float() * w).sum().backward()
    q, k, v = (qkv.detach()[:, :, i].transpose(1, 2).float().requires_grad_(True) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2)
    (ref * w).sum().backward()
    gref = torch.stack([t.grad.transpose(1, 2) for t in (q, k, v)], 2)
    e2 = (qkv.grad.float() - gref).abs().max().item() / max(1e-3, gref.abs().max().item())
    if e2 > 4e-2 or not torch.isfinite(qkv.grad.float()).all():
        bad += 1; print("MISMATCH", B, H, L, e2)
print("stress7 done, mismatches:", bad)
