"""The headline step (ViT-B/16 + BERT-base, batch 1024, both towers accelerated) eager against torch.compile(fullgraph=True,
backend="aot_eager"): every accelerated encoder and the loss are single operators in the traced graph (mmlearn_amd/compiled.py)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mmlearn_amd import ContrastiveLoss

dev = torch.device("cuda", 0)
b = int(os.environ.get("B", 1024))
batch = bench.synthetic_batch(b, 0, dev)
out = {"batch": b}
for mode in ("eager", "compiled"):
    torch._dynamo.reset()
    task = bench.build_task(ContrastiveLoss(), small=False, fused=True).to(dev)
    task.concurrent_encoders = False          # the traced step is one stream (the task switches its side streams off under compile anyway)
    opt = task.configure_optimizers()
    step_fn = torch.compile(task.training_step, backend="aot_eager", fullgraph=True) if mode == "compiled" else task.training_step

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = step_fn(batch, 0)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    out[mode + "_ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 2)
    out[mode + "_loss"] = round(float(loss.detach()), 5)
    del task, opt, step_fn
    torch.cuda.empty_cache()
print(json.dumps(out))
