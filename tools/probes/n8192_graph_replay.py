"""The N = 8192 loss path (forward + backward, both directions) eager against a HIP-graph replay of the same launches: how much of the
eager wall time is host enqueue (the matcher's host read, Python, ctypes) rather than device work."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmlearn_amd import ContrastiveLoss, LossPairSpec, _lib

dev = torch.device("cuda", 0)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 512
torch.manual_seed(0)
a = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
b = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
ids = torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1)
s = torch.tensor(1 / 0.07, device=dev, requires_grad=True)
fn, pairs = ContrastiveLoss(), [LossPairSpec(("rgb", "text"))]


def step(paired):
    a.grad = b.grad = s.grad = None
    loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs, fully_paired=paired)
    loss.float().backward()
    return loss


def timed(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


out = {"n": n, "eager_matcher_us": timed(lambda: step(None)), "eager_paired_hint_us": timed(lambda: step(True))}
ref = (float(step(True).detach().float()), a.grad.clone(), b.grad.clone(), s.grad.clone())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step(True)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
a.grad = b.grad = s.grad = None
with torch.cuda.graph(g):
    loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs, fully_paired=True)
    loss.float().backward()
out["graph_replay_us"] = timed(g.replay)
g.replay()
torch.cuda.synchronize()
out["replay_bit_identical"] = bool(float(loss.detach().float()) == ref[0] and torch.equal(a.grad, ref[1]) and torch.equal(b.grad, ref[2]) and torch.equal(s.grad, ref[3]))
_lib.profile_read(); _lib.profile_enable(True)
for _ in range(10):
    step(True)
torch.cuda.synchronize()
prof = _lib.profile_read(); _lib.profile_enable(False)
out["device_us"] = round(sum(v[1] for v in prof.values()) / 10 * 1e3, 1)
print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}))
