"""The token-0 query attention of the last layers (fused._cls_query_attention): F.scaled_dot_product_attention with ONE query per (sample,
head) against all keys, forward + backward, under each SDPA backend -- which one should serve it?"""
import sys, time, torch, torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel
dev = torch.device("cuda", 0)
for B, L in ((1024, 197), (1024, 77)):
    H, dh = 12, 64
    kv = torch.randn(B, L, 2, H, dh, device=dev, dtype=torch.bfloat16, requires_grad=True)
    q = torch.randn(B, H, 1, dh, device=dev, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, H, 1, dh, device=dev, dtype=torch.bfloat16)
    k, v = kv.unbind(2)
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            def run():
                with sdpa_kernel(be):
                    o = F.scaled_dot_product_attention(q, k.transpose(1, 2), v.transpose(1, 2), scale=0.125)
                o.backward(do)
                q.grad = None; kv.grad = None
            for _ in range(3): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize()
            print(B, L, name, round((time.perf_counter() - t0) / 10 * 1e6, 1), "us fwd+bwd")
        except Exception as e:
            print(B, L, name, "ERR", str(e)[:100])
