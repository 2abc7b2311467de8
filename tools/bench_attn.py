"""Attention microbenchmark at the ViT-B/16 shape of the headline workload (B=1024, H=12, L=197, dh=64, bf16)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib
from mmlearn_amd import kernels as K
from mmlearn_amd.attention import attention

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6

def main():
    B, H, L, dh = int(os.environ.get("B", 1024)), 12, int(os.environ.get("L", 197)), 64
    dev = torch.device("cuda", 0)
    mk = lambda: torch.randn(B, L, H * dh, device=dev).bfloat16().requires_grad_(True)
    q0, k0, v0 = mk(), mk(), mk()
    q, k, v = (x.view(B, L, H, dh).transpose(1, 2) for x in (q0, k0, v0))
    w = torch.randn(B, L, H, dh, device=dev).bfloat16()
    res = {"shape": [B, H, L, dh]}
    res["hip_fwd_us"] = round(t(lambda: K.attn_fwd(q, k, v, 0.125)), 1)
    res["sdpa_fwd_us"] = round(t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)), 1)
    def fb_hip():
        for x in (q0, k0, v0): x.grad = None
        attention(q, k, v, 0.125).backward(w)
    def fb_sdpa():
        for x in (q0, k0, v0): x.grad = None
        torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).backward(w)
    res["hip_fwd_bwd_us"] = round(t(fb_hip), 1)
    res["sdpa_fwd_bwd_us"] = round(t(fb_sdpa), 1)
    bytes_fwd = 4 * B * L * H * dh * 2
    res["hip_fwd_GBps"] = round(bytes_fwd / res["hip_fwd_us"] / 1e3, 1)
    print(json.dumps(res))

if __name__ == "__main__":
    main()
