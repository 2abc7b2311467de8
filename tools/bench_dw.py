"""Weight-gradient GEMM dW[N,K] = dY^T x: stock form (dY.t() @ x, both operands contraction-strided) vs a materialised
transpose dYt[N,M] @ x (the layout class of the fast dX kernels), incl. the transpose cost."""
import json, os, sys, time
import torch
dev = torch.device("cuda", 0)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for M in (1024 * 197, 1024 * 77):
    for N, K_ in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K_, device=dev).bfloat16()
        dyt = dy.t().contiguous()
        xt = x.t().contiguous()
        res = {"M": M, "N": N, "K": K_}
        res["stock_us"] = round(t(lambda: dy.t() @ x), 1)
        res["dyT_mm_us"] = round(t(lambda: dyt @ x), 1)
        res["both_T_us"] = round(t(lambda: dyt @ xt.t()), 1)
        res["dwT_xT_dy_us"] = round(t(lambda: xt @ dy), 1)
        res["transpose_dy_us"] = round(t(lambda: dy.t().contiguous()), 1)
        res["transpose_x_us"] = round(t(lambda: x.t().contiguous()), 1)
        print(json.dumps(res))
