"""Own persistent GEMM (csrc/mlp_gemm.hip, plain mode) against the tuned library on the 768 x 768 products of the step (attention
output projection forward and dX: the library's weakest shape, 841 TFLOP/s) and its neighbours."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmlearn_amd import kernels as K, tuned
dev = torch.device("cuda", 0)
tuned.enable()
def t(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for M in (201728, 78848):
    for N, Kd in ((768, 768), (2304, 768), (768, 2304)):
        a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16()
        lib = t(lambda: torch.nn.functional.linear(a, w))
        try:
            own = t(lambda: K.mlp_gemm_plain(a, w))
            ref = torch.nn.functional.linear(a, w).float(); got = K.mlp_gemm_plain(a, w).float()
            err = ((ref - got).abs().max() / ref.abs().max()).item()
        except Exception as e:
            own, err = None, str(e)[:80]
        print(M, N, Kd, "library", round(lib, 1), "own", own if own is None else round(own, 1), "TF lib", round(2 * M * N * Kd / lib / 1e6), "err", err)
