"""Recall@K: mmk_recall_ranks vs the reference's op sequence run on the same GPU (normalise, [b, M] scores
per batch, torch.topk, gather) -- the reference itself runs that sequence on the CPU."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K, _lib
from mmlearn_amd.ops import l2_normalize

dev = torch.device("cuda", 0)
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

for n, d, k in ((5000, 512, 10), (25000, 512, 10), (8192, 1024, 5)):
    g = torch.Generator().manual_seed(n)
    base = torch.randn(n, d, generator=g)
    x, y = (base + 0.7 * torch.randn(n, d, generator=g)).to(dev), base.to(dev)
    idx = torch.arange(n, device=dev)
    def hip():
        return (K.recall_ranks(l2_normalize(x), l2_normalize(y), idx) < k).float().mean()
    def eager(bs=1024):
        xn, yn = x / x.norm(dim=-1, keepdim=True), y / y.norm(dim=-1, keepdim=True)
        hits = []
        for s in range(0, n, bs):
            sc = xn[s:s + bs] @ yn.T
            pp = torch.zeros_like(sc, dtype=torch.bool)
            pp[torch.arange(sc.shape[0], device=dev), idx[s:s + bs]] = True
            hits.append(pp.gather(1, torch.topk(sc, k, dim=1)[1]).sum(1))
        return (torch.cat(hits) > 0).float().mean()
    r = {"n": n, "d": d, "k": k, "recall_hip": round(float(hip()), 5), "recall_eager": round(float(eager()), 5)}
    r["hip_ms"] = round(t(hip), 3); r["eager_gpu_ms"] = round(t(eager), 3)
    _lib.profile_enable(True); _lib.profile_read(); hip(); torch.cuda.synchronize()
    pr = _lib.profile_read(); _lib.profile_enable(False)
    r["count_pass_us"] = round(pr["recall_ranks"][1] / pr["recall_ranks"][0] * 1e3, 1)
    r["count_pass_TFs_f32"] = round(2.0 * n * n * d / (r["count_pass_us"] * 1e-6) / 1e12, 1)
    print(json.dumps(r))
