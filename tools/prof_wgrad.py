"""Driver for rocprofv3 --pmc passes over csrc/wgrad.hip at the encoder shapes (one shape per run: SHAPE=N,K  M=rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
M = int(os.environ.get("M", 1024 * 197))
N, K_ = (int(v) for v in os.environ.get("SHAPE", "768,768").split(","))
dev = torch.device("cuda", 0)
dy = torch.randn(M, N, device=dev).bfloat16()
x = torch.randn(M, K_, device=dev).bfloat16()
for _ in range(5):
    K.wgrad(dy, x)
torch.cuda.synchronize()
