"""Host-side cost of one loss forward + backward at the headline shape (N = 1024, D = 512): wall time per call with the GPU kept
busy-free (sync each iteration) and free-running, for the one-launch path and the tiled path, with and without the
`fully_paired` hint (no matcher, no status read-back).    python tools/host_loss_wall.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import ContrastiveLoss, LossPairSpec, kernels as K

dev = torch.device("cuda", 0)
n, d = int(os.environ.get("N", 1024)), int(os.environ.get("D", 512))
torch.manual_seed(0)
mk = lambda: torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).requires_grad_(True)
a, b = mk(), mk()
ids = torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1)
s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
pairs = [LossPairSpec(("rgb", "text"))]
fn = ContrastiveLoss()

def step(hint):
    a.grad = b.grad = s.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs, fully_paired=hint)
    loss.backward()

out = {"n": n, "d": d}
K.FUSED_LOSS = True
for _ in range(2000):   # clocks, allocator and the launch queue in steady state before the first measured configuration
    step(True)
torch.cuda.synchronize()
for path in ("one_launch", "tiled"):
    K.FUSED_LOSS = path == "one_launch"
    for hint in (True, None):
        for _ in range(20):
            step(hint)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            step(hint)
        torch.cuda.synchronize()
        free = (time.perf_counter() - t0) / 300 * 1e6
        t0 = time.perf_counter()
        for _ in range(300):
            step(hint)
            torch.cuda.synchronize()
        synced = (time.perf_counter() - t0) / 300 * 1e6
        out[f"{path}{'_hint' if hint else ''}"] = {"free_running_us": round(free, 1), "synced_us": round(synced, 1)}
out["workspaces_allocated"] = {str(k[1:]): p.allocs for k, p in K._FUSED_PLANS.items()}
print(json.dumps(out))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    K.FUSED_LOSS = True
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        step(True)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
