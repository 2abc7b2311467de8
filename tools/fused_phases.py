"""Where the workgroups of the one-launch loss (csrc/clip_fused.hip) spend their time: realtime-clock stamps per phase.
Needs the debug-switch build:
    make -C mmlearn_amd/csrc VARIANT=_dbg EXTRA=-DMMK_DEBUG_SWITCHES -j8
    MMK_LIB_VARIANT=_dbg python tools/fused_phases.py [N D dtype]
Stamps (10 ns units, relative to the earliest workgroup start): 0 start, 1 S tile done, 2 statistics published, 3 strip statistics
arrived, 4 G stored, 5 G strip arrived (first gradient job), 6 gradient jobs done, 7 ticket drawn."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib, kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dt = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).to(dt)
b = torch.nn.functional.normalize(torch.randn(n, d, device=dev), dim=-1).to(dt)
s = torch.tensor([1 / 0.07], device=dev)
plan = K.clip_fused_plan(dev, [n], d, dt)
grid = plan.grid
stamps = torch.zeros(grid * 16, dtype=torch.int64, device=dev)
rows = []
for it in range(25):
    if it == 5:
        _lib.check(_lib.lib().mmk_clip_fused_debug_stamps(stamps.data_ptr()))
    loss, run = K.clip_fused_forward(plan, [(a, b, None, None, n, 1.0)], d, s, True)
    torch.cuda.synchronize()
    run.release()
    if it >= 5:
        st = stamps.cpu().numpy().reshape(grid, 16).astype(np.int64)
        rows.append(st - st[:, 0].min())
_lib.check(_lib.lib().mmk_clip_fused_debug_stamps(None))
st = np.median(np.stack(rows), axis=0) / 100.0   # us
names = ["start", "S tile", "stats out", "stats in", "G out", "G in", "grads", "ticket"]
out = {"n": n, "d": d, "dtype": str(dt), "grid": grid}
for k, nm in enumerate(names):
    col = st[:, k]
    out[nm] = {"min": round(float(col.min()), 2), "median": round(float(np.median(col)), 2), "max": round(float(col.max()), 2)}
seg = np.diff(st[:, :8], axis=1)
out["segments_median_us"] = {f"{names[k]}->{names[k + 1]}": round(float(np.median(seg[:, k])), 2) for k in range(7)}
# finer stamps (thread 0 of every workgroup): 8 addresses ready, 9 S-tile loads issued, 10 first stage in LDS, 11 LSEs merged,
# 12 G stores issued, 13 first gradient chunk in LDS, 14 gradient loop done
fine = {"setup": (0, 8), "issue S loads": (8, 9), "first S stage landed": (9, 10), "rest of S tile": (10, 1), "spin1 exit -> LSEs merged": (3, 11),
        "LSEs -> G tile computed": (11, 15), "G computed -> G stores issued": (15, 12), "G stores drained + counters": (12, 4), "G arrived -> first chunk in LDS": (5, 13),
        "gradient loop": (13, 14), "partial-tile sum + stores": (14, 6)}
out["fine_median_us"] = {k: round(float(np.median(st[:, b] - st[:, a_])), 2) for k, (a_, b) in fine.items()}
print(json.dumps(out))
