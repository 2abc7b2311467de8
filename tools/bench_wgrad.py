"""Weight-gradient GEMM: csrc/wgrad.hip vs hipBLASLt (dY.t() @ x) at the encoder shapes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
from mmlearn_amd import _lib
dev = torch.device("cuda", 0)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
SHAPES = [(M, N, K_) for M in (1024 * 197, 1024 * 77) for N, K_ in ((768, 768), (768, 3072), (3072, 768), (2304, 768))]
if os.environ.get("SHAPES") == "ijepa":   # ViT-L/16 context / target rows and the 384-wide predictor (tools/bench_ijepa_step.py)
    SHAPES = [(M, N, K_) for M in (4096, 8192, 12288, 25088) for N, K_ in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096))]
    SHAPES += [(M, N, K_) for M in (16384, 53760) for N, K_ in ((384, 384), (1152, 384), (1536, 384), (384, 1536))]
if os.environ.get("SHAPES") == "htsat":   # HTSAT at batch 256: four resolutions, packed q|k|v, out, fc1, fc2
    SHAPES = [(256 * t, n * c, k * c) for t, c in ((4096, 96), (1024, 192), (256, 384), (64, 768)) for n, k in ((3, 1), (1, 1), (4, 1), (1, 4))]
for M, N, K_ in SHAPES:
    if True:
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K_, device=dev).bfloat16()
        res = {"M": M, "N": N, "K": K_, "GF": round(2 * M * N * K_ / 1e9)}
        res["hipblaslt_us"] = round(t(lambda: dy.t() @ x), 1)
        res["wgrad_us"] = round(t(lambda: K.wgrad(dy, x)), 1)
        _lib.profile_enable(True); _lib.profile_read()
        for _ in range(5): K.wgrad(dy, x)
        torch.cuda.synchronize()
        pr = _lib.profile_read(); _lib.profile_enable(False)
        res["wgrad_main_kernel_us"] = round(pr["wgrad"][1] / pr["wgrad"][0] * 1e3, 1)
        res["wgrad_TFs"] = round(2 * M * N * K_ / res["wgrad_us"] / 1e6, 1)
        ref = dy.t() @ x
        res["max_rel_err"] = float(((K.wgrad(dy, x) - ref.float()).abs().max() / ref.float().abs().max()).item())
        print(json.dumps(res))
