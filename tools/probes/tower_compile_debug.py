import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mmlearn_amd import ContrastiveLoss
dev = torch.device("cuda", 0)
batch = bench.synthetic_batch(16, 0, dev, padded=True)
def make():
    task = bench.build_task(ContrastiveLoss(), small=True, fused=True).to(dev)
    task.concurrent_encoders = False
    for m in task.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        cfg = getattr(m, "config", None)
        if cfg is not None and hasattr(cfg, "attention_probs_dropout_prob"): cfg.attention_probs_dropout_prob = 0.0
    return task, task.configure_optimizers()
def run(compiled):
    torch._dynamo.reset()
    task, opt = make()
    step = torch.compile(task.training_step, backend="aot_eager", fullgraph=True) if compiled else task.training_step
    out = []
    for i in range(3):
        opt.zero_grad(set_to_none=False)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = step(batch, 0)
        loss.backward()
        gn = {n: float(p.grad.float().norm()) for n, p in list(task.named_parameters())[:3] + list(task.named_parameters())[-3:]}
        opt.step()
        out.append((float(loss.detach()), gn))
    return out
a, b, c = run(False), run(False), run(True)
for i in range(3):
    print(i, a[i][0], b[i][0], c[i][0])
print(a[0][1]); print(c[0][1])
