import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
torch.manual_seed(0)
dev = torch.device("cuda", 0)
for (B, H, L) in [(1, 1, 197), (3, 4, 197), (2, 12, 77), (600, 2, 197)]:
    q, k, v = (torch.randn(B, L, H * 64, device=dev).bfloat16().view(B, L, H, 64).transpose(1, 2) for _ in range(3))
    o, lse = K.attn_fwd(q, k, v, 0.125)
    s = (q.float() @ k.float().transpose(-1, -2)) * 0.125
    ref = (torch.softmax(s, -1) @ v.float()).transpose(1, 2)
    err = (o.float() - ref).abs().amax(dim=(1, 2, 3))
    print(B, H, L, "fwd max err per b:", err[:6].tolist(), "lse err", (lse - torch.logsumexp(s, -1)).abs().max().item())
    bad = (o.float() - ref).abs().amax(dim=3)  # [B, L, H]
    idx = (bad > 0.05).nonzero()
    print("  bad rows:", idx[:10].tolist(), len(idx))
