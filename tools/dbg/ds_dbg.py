import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmlearn_amd import ContrastiveLoss, LossPairSpec
from mmlearn_amd import kernels as K
import mmlearn_amd.losses as L
dev = torch.device("cuda", 0)
n, d, scale = 700, 512, 1/0.07
g = torch.Generator().manual_seed(n + d)
a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).bfloat16()
b = torch.nn.functional.normalize(0.6 * a.float() + 0.8 * torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1), dim=-1).bfloat16()
ids = torch.stack([torch.zeros(n, dtype=torch.long), torch.arange(n)], 1).to(dev)
orig = L._Run._fused_backward
def dbg(self, grad_out):
    run = self.fused
    torch.cuda.synchronize()
    print("before bwd: ds_raw", run.ds_raw.tolist(), "ds_acc", run.ds_acc.tolist(), "grad_out", grad_out, flush=True)
    r = orig(self, grad_out)
    torch.cuda.synchronize()
    print("after bwd: ds", r[0], flush=True)
    return r
L._Run._fused_backward = dbg
for it in range(4):
    ea, eb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    s = torch.tensor(scale, device=dev, requires_grad=True)
    loss = ContrastiveLoss()({"rgb_embedding": ea, "text_embedding": eb}, {"rgb": ids, "text": ids}, s, [LossPairSpec(("rgb", "text"))])
    loss.float().backward()
    print(it, float(loss), float(s.grad), flush=True)
