"""Driver for rocprofv3 --pmc passes over the rank-of-positive kernel (csrc/metrics.hip): N = M = 25,000, D = 512, 3 launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
from mmlearn_amd.ops import l2_normalize
n, d = int(os.environ.get("N", 25000)), int(os.environ.get("D", 512))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
base = torch.randn(n, d, generator=g)
x, y = l2_normalize((base + 0.7 * torch.randn(n, d, generator=g)).to(dev)), l2_normalize(base.to(dev))
idx = torch.arange(n, device=dev)
for _ in range(3):
    K.recall_ranks(x, y, idx)
torch.cuda.synchronize()
