"""Where the HOST time of one rank's share of the row-sharded loss goes (R = 1024 x C = 8192, both directions): cProfile of the
step of bench.py's `loss_shard` leg, plus wall per step with and without a device sync per step (host-bound vs device-bound).

    python tools/profile_shard_host.py [--iters 200] [--top 40]
"""
import argparse, cProfile, json, os, pstats, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib, kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    R, C, D, p0 = 1024, 8192, 512, 3 * 1024
    torch.manual_seed(0)
    A = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    B = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
    scale = torch.tensor([1 / 0.07], device=dev)
    upstream = torch.ones((), device=dev)
    comp = _lib.COMPUTE_BF16
    kg = 1.0 / (2.0 * C)
    marks = {}

    def step(timed=False):
        t = [time.perf_counter()]
        (ag, agt), (bg, bgt) = K.pack_rows_many([(A, None, C, False, True), (B, None, C, False, True)], comp)
        t.append(time.perf_counter())
        dirs = []
        for x, y, yt in ((K.slice_packed(ag, p0), bg, bgt), (K.slice_packed(bg, p0), ag, agt)):
            dirs.append(K.Direction(x=x, y=y, y_t=yt, r=R, c=C, label_off=p0, kappa=kg, ds_kappa=kg))
        dirs[1].s_row = dirs[1].s_col = dirs[1].s_diag = 0.0
        K.clip_forward(dirs, D, comp, scale)
        t.append(time.perf_counter())
        for dr, other in ((dirs[0], dirs[1]), (dirs[1], dirs[0])):
            dr.lse_col = other.lse.repeat(C // R).contiguous()
            dr.dx = torch.zeros((R, D), dtype=torch.bfloat16, device=dev)
        ds = torch.zeros(1, device=dev)
        t.append(time.perf_counter())
        K.clip_backward(dirs, D, comp, scale, upstream, ds)
        t.append(time.perf_counter())
        if timed:
            for k, name in enumerate(("pack", "forward", "glue", "backward")):
                marks[name] = marks.get(name, 0.0) + (t[k + 1] - t[k])

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step(True)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
        torch.cuda.synchronize()
    wall_sync = time.perf_counter() - t0
    out = {"wall_us": round(wall / a.iters * 1e6, 1), "enqueue_us": round(t_enq / a.iters * 1e6, 1),
           "wall_us_with_sync_per_step": round(wall_sync / a.iters * 1e6, 1),
           "host_us_by_phase": {k: round(v / a.iters * 1e6, 1) for k, v in marks.items()}}
    print(json.dumps(out))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
