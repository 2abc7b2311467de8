"""Loss-path microbenchmark (single GPU): per-kernel HIP-event times of ContrastiveLoss fwd+bwd.

    python tools/bench_loss.py [--n 1024 --d 512 --dtype bf16 --iters 50]

Prints one JSON line per shape with the per-kernel average durations (from the library's HIP-event
recorder, i.e. on the launch stream) and executed / algorithmic TFLOP/s of the three MFMA kernels.
"""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mmlearn_amd import ContrastiveLoss, LossPairSpec, _lib  # noqa: E402


def run(n, d, dtype, iters, warmup=5, mods=2, path="auto", autocast=False):
    from mmlearn_amd import kernels as K

    K.FUSED_LOSS = path != "tiled"
    dev = torch.device("cuda", 0)
    tdt = {"bf16": torch.bfloat16, "fp32": torch.float32}[dtype]
    torch.manual_seed(0)
    names = ["rgb", "text", "audio"][:mods]
    embs = {f"{m}_embedding": torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).to(tdt).requires_grad_(True) for m in names}
    ids = {m: torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1) for m in names}
    s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
    pairs = [LossPairSpec((a, b)) for i, a in enumerate(names) for b in names[i + 1:]]
    fn = ContrastiveLoss()

    def step():
        for t in embs.values():
            t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss = fn(embs, ids, s, pairs)
        loss.float().backward()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / iters
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    # un-profiled wall (the recorder adds event records)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    wall_ms_np = (time.perf_counter() - t0) * 1e3 / iters
    n_pairs = len(pairs)
    per = {k: round(ms / cnt * 1e3, 2) for k, (cnt, ms) in prof.items()}  # us per launch
    gemm = 2.0 * n * n * d * n_pairs  # one [N,N,D] product per pair
    out = {"n": n, "d": d, "dtype": dtype + ("+autocast" if autocast else ""), "path": path, "pairs": n_pairs, "wall_ms_profiled": round(wall_ms, 3), "wall_ms": round(wall_ms_np, 3),
           "kernel_us": per, "device_us_total": round(sum(ms for _, ms in prof.values()) / iters * 1e3, 1)}
    for k, executed in (("sim_stats", 2 * gemm), ("sim_grad", 2 * gemm), ("grad_gemm", 2 * gemm)):
        if k in per:
            out[f"{k}_exec_tflops"] = round(executed / (per[k] * 1e-6) / 1e12, 1)
    out["algorithmic_tflops_fwd_bwd"] = round(3 * gemm / (out["device_us_total"] * 1e-6) / 1e12, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="*", default=[1024, 4096, 8192])
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--mods", type=int, default=2)
    ap.add_argument("--path", nargs="*", default=["auto"], help="auto (one resident-grid launch where it applies) | tiled")
    ap.add_argument("--autocast", action="store_true", help="f32 embeddings under bf16 autocast (what the task hands the loss)")
    a = ap.parse_args()
    for n in a.n:
        for path in a.path:
            run(n, a.d, a.dtype, a.iters, mods=a.mods, path=path, autocast=a.autocast)
