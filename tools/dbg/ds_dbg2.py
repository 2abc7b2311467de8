import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from mmlearn_amd import ContrastiveLoss, LossPairSpec
from mmlearn_amd import kernels as K
import mmlearn_amd.losses as L
import test_clip_gpu as T
dev = torch.device("cuda", 0)
T.test_no_grad_and_eval_paths()
rec = []
orig_b = K.clip_fused_backward
def dbg_b(run, grads, scale, upstream, dscale):
    before = (run.ds_raw.clone(), None if dscale is None else dscale.clone(), upstream.clone(), scale.clone())
    orig_b(run, grads, scale, upstream, dscale)
    after = None if dscale is None else dscale.clone()
    rec.append((before, after))
K.clip_fused_backward = dbg_b
n, d, scale = 700, 512, 1/0.07
g = torch.Generator().manual_seed(n + d)
a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).bfloat16()
b = torch.nn.functional.normalize(0.6 * a.float() + 0.8 * torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1), dim=-1).bfloat16()
ids = torch.stack([torch.zeros(n, dtype=torch.long), torch.arange(n)], 1).to(dev)
res = []
for it in range(4):
    ea, eb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    s = torch.tensor(scale, device=dev, requires_grad=True)
    loss = ContrastiveLoss()({"rgb_embedding": ea, "text_embedding": eb}, {"rgb": ids, "text": ids}, s, [LossPairSpec(("rgb", "text"))])
    loss.float().backward()
    res.append((float(loss.detach()), float(s.grad)))
torch.cuda.synchronize()
for (bf, af), r in zip(rec, res):
    print("ds_raw", bf[0].tolist(), "acc_before", bf[1].tolist(), "upstream", bf[2].tolist(), "scale", bf[3].tolist(), "acc_after", af.tolist(), "->", r)
