"""hipBLASLt timing of the encoder projections: separate q/k/v Linears vs one fused [E -> 3E] Linear (fwd and bwd)."""
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
    E = 768
    x = torch.randn(M, E, device=dev).bfloat16().requires_grad_(True)
    ws = [torch.randn(E, E, device=dev).bfloat16().requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(E, device=dev).bfloat16().requires_grad_(True) for _ in range(3)]
    wc = torch.cat([w.detach() for w in ws], 0).requires_grad_(True)
    bc = torch.cat([b.detach() for b in bs], 0).requires_grad_(True)
    res = {"M": M}
    res["fwd_3x_us"] = round(t(lambda: [F.linear(x, w, b) for w, b in zip(ws, bs)]), 1)
    res["fwd_fused_us"] = round(t(lambda: F.linear(x, wc, bc)), 1)
    g = torch.randn(M, E, device=dev).bfloat16()
    g3 = torch.randn(M, 3 * E, device=dev).bfloat16()
    def bwd3():
        x.grad = None
        outs = [F.linear(x, w, b) for w, b in zip(ws, bs)]
        torch.autograd.backward(outs, [g, g, g])
    def bwdf():
        x.grad = None
        F.linear(x, wc, bc).backward(g3)
    res["fwdbwd_3x_us"] = round(t(bwd3), 1)
    res["fwdbwd_fused_us"] = round(t(bwdf), 1)
    print(json.dumps(res))
