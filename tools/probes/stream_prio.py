"""Headline step with the first tower (ViT) on a high-priority HIP stream, the second (BERT) on the task's side stream:
does giving the long tower priority let the short one fill its tails all the way through the step?"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from mmlearn_amd import ContrastiveLoss, tuned

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tuned.enable()
print("priority range", torch.cuda.Stream.priority_range())
task = bench.build_task(ContrastiveLoss(static_shapes=True), False, fused=True).to(dev)
opt = task.configure_optimizers()
batch = bench.synthetic_batch(1024, 0, dev)

def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss

res = {}
res["default"] = round(bench._timed_steps(step, 3, 8) * 1e3, 2)
lo, hi = torch.cuda.Stream.priority_range()
hp = torch.cuda.Stream(priority=hi)
hp.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(hp):
    res["main_high_priority"] = round(bench._timed_steps(step, 3, 8) * 1e3, 2)
torch.cuda.current_stream().wait_stream(hp)
res["default_again"] = round(bench._timed_steps(step, 2, 8) * 1e3, 2)
with torch.cuda.stream(hp):
    res["main_high_priority_again"] = round(bench._timed_steps(step, 2, 8) * 1e3, 2)
print(json.dumps(res))
