"""How far ahead of the GPU is the host?  Times the bench step's Python/launch side (no synchronisation inside the loop)
against the synchronised step time; host_ms well below step_ms means the step is GPU-bound and launch gaps are hidden."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmlearn_amd import ContrastiveLoss

dev = torch.device("cuda", 0)
task = bench.build_task(ContrastiveLoss(static_shapes=True), small=False, fused=True).to(dev)
opt = task.configure_optimizers()
batch = bench.synthetic_batch(1024, 0, dev)

def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task.training_step(batch, 0)
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(6):
    h0 = time.perf_counter()
    step()
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) * 1e3 / 6
# the same with a drain before every step: the host's own cost per step when it never has to wait for queue space
solo = []
for _ in range(4):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    step()
    solo.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
print(json.dumps({"step_ms": round(total, 1), "host_ms_per_step_pipelined": [round(h, 1) for h in host],
                  "host_ms_per_step_after_drain": [round(h, 1) for h in solo]}))
