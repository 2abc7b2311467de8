"""cProfile of the Python side of one loss fwd+bwd (N=1024) to find host overhead."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import ContrastiveLoss, LossPairSpec
dev = torch.device("cuda", 0)
n, d = 1024, 512
a = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
b = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).bfloat16().requires_grad_(True)
ids = torch.stack([torch.zeros(n, dtype=torch.long, device=dev), torch.arange(n, device=dev)], 1)
s = torch.tensor(14.0, device=dev, requires_grad=True)
fn = ContrastiveLoss(); pairs = [LossPairSpec(("rgb", "text"))]
def step():
    a.grad = None; b.grad = None
    loss = fn({"rgb_embedding": a, "text_embedding": b}, {"rgb": ids, "text": ids}, s, pairs)
    loss.float().backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
