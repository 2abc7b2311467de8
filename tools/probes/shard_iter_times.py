import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from mmlearn_amd import _lib, kernels as K
dev = torch.device("cuda", 0)
R, C, D = 1024, 8192, 512
p0 = 3 * R
torch.manual_seed(0)
A = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
B = torch.nn.functional.normalize(torch.randn(C, D, device=dev), dim=-1).bfloat16()
scale = torch.tensor([1 / 0.07], device=dev)
upstream = torch.ones((), device=dev)
comp = _lib.COMPUTE_BF16
kg = 1.0 / (2.0 * C)
def step():
    (ag, agt), (bg, bgt) = K.pack_rows_many([(A, None, C, False, True), (B, None, C, False, True)], comp)
    dirs = []
    for x, y, yt in ((K.slice_packed(ag, p0), bg, bgt), (K.slice_packed(bg, p0), ag, agt)):
        dirs.append(K.Direction(x=x, y=y, y_t=yt, r=R, c=C, label_off=p0, kappa=kg, ds_kappa=kg))
    dirs[1].s_row = dirs[1].s_col = dirs[1].s_diag = 0.0
    K.clip_forward(dirs, D, comp, scale)
    for dr, other in ((dirs[0], dirs[1]), (dirs[1], dirs[0])):
        dr.lse_col = other.lse.repeat(C // R).contiguous()
        dr.dx = torch.zeros((R, D), dtype=torch.bfloat16, device=dev)
    ds = torch.zeros(1, device=dev)
    K.clip_backward(dirs, D, comp, scale, upstream, ds)
for _ in range(5):
    step()
torch.cuda.synchronize()
_lib.profile_read()
_lib.profile_enable(True)
tp = []
for _ in range(20):
    t0 = time.perf_counter(); step(); tp.append(round((time.perf_counter() - t0) * 1e6))
torch.cuda.synchronize()
t0 = time.perf_counter()
prof = _lib.profile_read()
t_read = time.perf_counter() - t0
_lib.profile_enable(False)
ts = []
t00 = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter(); step(); ts.append(round((time.perf_counter() - t0) * 1e6))
t_enq = time.perf_counter() - t00
torch.cuda.synchronize()
wall = (time.perf_counter() - t00) / 20
print(json.dumps({"profiled_iter_us": tp, "profile_read_ms": round(t_read * 1e3, 2), "timed_iter_us": ts, "enqueue_us": round(t_enq / 20 * 1e6, 1), "wall_us": round(wall * 1e6, 1),
                  "device_us": round(sum(v[1] for v in prof.values()) / 20 * 1e3, 1)}))
