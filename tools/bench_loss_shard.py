"""One rank's share of the row-sharded loss path (BASELINE configs[2]: W = 8, per-rank batch 1024, global batch 8192) on one
GPU: the rank's two directions S_r = A_r B_all^T and T_r = B_r A_all^T (R = 1024 rows, C = 8192 columns, label_off = r R),
forward statistics + merge, gradient tiles, gradient GEMMs, finalize -- the launches mmlearn_amd.losses issues between the
all-gather and the LSE all-reduce, and after it (the collectives themselves are not part of this tool).

    python tools/bench_loss_shard.py [--rows 1024 --cols 8192 --d 512 --rank 3 --iters 30]

Per-kernel HIP-event times; algorithmic work per rank = 8 R C D (2 R C D per direction forward, the same
backward).  The column LSEs that the real path receives from the all-reduce are taken from a single-GPU full-batch run."""
import argparse, json, os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib, kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--cols", type=int, default=8192)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    R, C, D = a.rows, a.cols, a.d
    p0 = a.rank * R
    assert p0 + R <= C
    torch.manual_seed(0)
    A = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    B = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    scale = torch.tensor([1 / 0.07], device=dev)
    upstream = torch.ones((), device=dev)
    comp = _lib.COMPUTE_BF16
    kg = 1.0 / (2.0 * C)

    want_t = not K.backward_recomputes_on_chip(R, C, D, comp, 2)   # as mmlearn_amd.losses decides: no transposed copies for the one-kernel backward

    def step():
        (ag, agt), (bg, bgt) = K.pack_rows_many([(A, None, C, False, want_t), (B, None, C, False, want_t)], comp)
        dirs = []
        for x, y, yt in ((K.slice_packed(ag, p0), bg, bgt), (K.slice_packed(bg, p0), ag, agt)):
            dirs.append(K.Direction(x=x, y=y, y_t=yt, r=R, c=C, label_off=p0, kappa=kg, ds_kappa=kg))
        dirs[1].s_row = dirs[1].s_col = dirs[1].s_diag = 0.0
        K.clip_forward(dirs, D, comp, scale)
        # stand-in for the all-reduced column LSEs: this rank's own rows are real, the other ranks' entries reuse them
        for dr, other in ((dirs[0], dirs[1]), (dirs[1], dirs[0])):
            dr.lse_col = other.lse.repeat(C // R).contiguous()
            dr.dx = torch.zeros((R, D), dtype=torch.bfloat16, device=dev)
        ds = torch.zeros(1, device=dev)
        K.clip_backward(dirs, D, comp, scale, upstream, ds)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    per = {k: round(ms / cnt * 1e3, 2) for k, (cnt, ms) in prof.items() if cnt}
    dev_us = sum(ms for _, ms in prof.values()) / a.iters * 1e3
    flops = 8.0 * R * C * D
    out = {"rows": R, "cols": C, "d": D, "rank": a.rank, "kernel_us": per, "device_us_total": round(dev_us, 1), "wall_us": round(wall * 1e6, 1),
           "algorithmic_gflop_per_rank": round(flops / 1e9, 1), "algorithmic_tflops": round(flops / (dev_us * 1e-6) / 1e12, 1),
           "sim_stats_algorithmic_tflops": round(4.0 * R * C * D / (per["sim_stats"] * 1e-6) / 1e12, 1) if "sim_stats" in per else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
