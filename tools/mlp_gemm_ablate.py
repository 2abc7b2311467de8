"""Timing ablations of csrc/mlp_gemm.hip (debug-switch build only; results are WRONG by design in every arm but 0):
MMK_MLP_GEMM_DBG bits: 4 = no C stores, 8 = no epilogue arithmetic, 64 = every LDS-DMA re-reads one cached 64 KiB, 512 = no PIPE.
    MMK_LIB_VARIANT=_dbg python tools/mlp_gemm_ablate.py"""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K

dev = torch.device("cuda", 0)
M, E, H = 1024 * 197, 768, 3072
dy = torch.randn(M, E, device=dev).bfloat16()
w2t = (torch.randn(H, E, device=dev) / 55).bfloat16()
g = torch.rand(M, H, device=dev).bfloat16()
arms = {"plain": lambda: K.mlp_gemm_plain(dy, w2t), "bwd_mul": lambda: K.mlp_gemm_bwd_mul(dy, w2t, g)}
out = {}
for bits in (0, 4, 64, 68, 512, 516):
    os.environ["MMK_MLP_GEMM_DBG"] = str(bits)
    for name, fn in arms.items():
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 100)
        out[f"{name}@{bits}"] = round(statistics.median(ts), 1)
print(json.dumps(out))
