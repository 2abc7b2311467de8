"""Backward-only timing of the attention kernels (events around K.attn_bwd)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
B, H, L = int(os.environ.get("B", 1024)), 12, int(os.environ.get("L", 197))
p = float(os.environ.get("P", 0.0))
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(B, L, H * 64, device=dev).bfloat16().view(B, L, H, 64).transpose(1, 2) for _ in range(3))
do = torch.randn(B, L, H, 64, device=dev).bfloat16()
o, lse = K.attn_fwd(q, k, v, 0.125, p, 7)
for _ in range(3): K.attn_bwd(q, k, v, o, lse, do, 0.125, p, 7)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): K.attn_bwd(q, k, v, o, lse, do, 0.125, p, 7)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"L": L, "p": p, "seven_product": os.environ.get("MMK_ATTN_BWD7") is not None, "bwd_us": round(e0.elapsed_time(e1) * 100, 1)}))
