"""Driver for a rocprofv3 --pmc pass over the library's forward GEMMs at the ViT-B/16 MLP shapes (compare with tools/prof_wgrad.py)."""
import torch
dev = torch.device("cuda", 0)
M = 1024 * 197
for n, k in ((3072, 768), (768, 3072)):
    x = torch.randn(M, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    for _ in range(5):
        torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
