"""hipBLASLt timing of the encoder Linear shapes under the two weight layouts PyTorch can hand it:
W stored [N, K] (nn.Linear's layout; forward is a 'TN' GEMM) vs W^T stored [K, N] and passed as a transposed view
(forward becomes 'NN').  Forward-only and forward+backward, bf16, with bias."""
import json, os, sys, time
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6

for M in (1024 * 197, 1024 * 77):
    for K_, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
        x = torch.randn(M, K_, device=dev).bfloat16().requires_grad_(True)
        w = torch.randn(N, K_, device=dev).bfloat16().requires_grad_(True)           # [N, K]
        wt = w.detach().t().contiguous().requires_grad_(True)                          # [K, N]
        b = torch.randn(N, device=dev).bfloat16().requires_grad_(True)
        g = torch.randn(M, N, device=dev).bfloat16()
        res = {"M": M, "K": K_, "N": N, "GF": round(2 * M * K_ * N / 1e9)}
        res["fwd_NK_us"] = round(t(lambda: F.linear(x, w, b)), 1)
        res["fwd_KN_us"] = round(t(lambda: F.linear(x, wt.t(), b)), 1)
        res["fwd_KN_addmm_us"] = round(t(lambda: torch.addmm(b, x, wt)), 1)
        def fb(wv):
            x.grad = None
            F.linear(x, wv, b).backward(g)
        res["fb_NK_us"] = round(t(lambda: fb(w)), 1)
        res["fb_KN_us"] = round(t(lambda: fb(wt.t())), 1)
        print(json.dumps(res))
