"""The headline step (configs[1], batch 1024) eagerly on one stream / on two streams, and captured into ONE HIP graph (single stream)
and replayed: what do ~1,100 launches per step cost on the device timeline when nothing waits for the host?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from functools import partial
import bench
from mmlearn_amd import ContrastiveLoss, tuned
from mmlearn_amd.optim import AdamW

dev = torch.device("cuda", 0)
tuned.enable()
out = {}


def build(streams):
    task = bench.build_task(ContrastiveLoss(static_shapes=True), False, fused=True).to(dev)
    task.concurrent_encoders = streams
    if not streams:
        task.match_ahead = False
    task.optimizer = partial(AdamW, lr=1e-4, weight_decay=0.1, capturable=True)
    opt = task.configure_optimizers()
    return task, (opt["optimizer"] if isinstance(opt, dict) else opt)


batch = bench.synthetic_batch(1024, 0, dev)
if os.environ.get("PAIRED"):
    batch["fully_paired"] = True


def make_step(task, opt):
    def step():
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step(batch, 0)
        loss.backward()
        opt.step()
        return loss
    return step


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for name, streams in (("eager_one_stream", False), ("eager_two_streams", True)):
    task, opt = build(streams)
    step = make_step(task, opt)
    for _ in range(3): step()
    out[name + "_ms"] = round(timed(step, 6), 2)
    del task, opt, step
    torch.cuda.empty_cache()
task, opt = build(False)
step = make_step(task, opt)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(2): g.replay()
out["graph_one_stream_ms"] = round(timed(g.replay, 6), 2)
out["loss"] = float(loss.detach())
out["peak_gib"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
print(json.dumps(out))
