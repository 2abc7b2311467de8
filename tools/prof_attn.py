"""Tiny driver for rocprofv3 runs of the attention kernels alone (ViT-B/16 shape by default)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K

B, H, L = int(os.environ.get("B", 1024)), 12, int(os.environ.get("L", 197))
p = float(os.environ.get("P", 0.0))
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(B, L, H * 64, device=dev).bfloat16().view(B, L, H, 64).transpose(1, 2) for _ in range(3))
do = torch.randn(B, L, H, 64, device=dev).bfloat16()
for _ in range(int(os.environ.get("IT", 5))):
    o, lse = K.attn_fwd(q, k, v, 0.125, p, 7)
    K.attn_bwd(q, k, v, o, lse, do, 0.125, p, 7)
torch.cuda.synchronize()
