"""Windowed attention forward + backward at ONE shape, for rocprofv3 passes (kernel trace, or --pmc FETCH_SIZE / WRITE_SIZE):
    rocprofv3 --pmc FETCH_SIZE -d out -o f --output-format csv -- python3 tools/prof_window_attn.py [--nw 64 --heads 4 --batch 256 --iters 5]
and, given the two counter files, the HBM-side bytes per launch against the algorithmic bytes:
    python tools/prof_window_attn.py --summarise fetch_counter_collection.csv write_counter_collection.csv [--nw ...]"""
import argparse, collections, csv, json, math, os, sys

ap = argparse.ArgumentParser()
ap.add_argument("--nw", type=int, default=64); ap.add_argument("--heads", type=int, default=4); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--summarise", nargs=2, default=None)
a = ap.parse_args()
dh = 24
C, Bw = a.heads * dh, a.batch * a.nw
unit = Bw * 64 * C * 2
lse = Bw * a.heads * 64 * 4
if a.summarise:
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/prof_window_attn.py; HBM-side bytes per launch = 2*FETCH_SIZE + WRITE_SIZE "
                   "(KiB counters; gfx950 counts 128-byte requests at 64)", "windows_per_sample": a.nw, "heads": a.heads, "batch": a.batch, "masked": True}
    per = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    for path in a.summarise:
        for r in csv.DictReader(open(path)):
            if "win_attn" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                e = per["bwd" if "bwd" in r["Kernel_Name"] else "fwd"][r["Counter_Name"]]
                e[0] += float(r["Counter_Value"]); e[1] += 1
    alg = {"fwd": 4 * unit + lse, "bwd": 7 * unit + lse}
    for k, v in per.items():
        f, w = v["FETCH_SIZE"][0] / max(1, v["FETCH_SIZE"][1]), v["WRITE_SIZE"][0] / max(1, v["WRITE_SIZE"][1])
        hbm = int((2 * f + w) * 1024)
        out[k] = {"launches_profiled": v["FETCH_SIZE"][1], "fetch_kib": round(f, 1), "write_kib": round(w, 1), "hbm_bytes_per_launch": hbm,
                  "algorithmic_bytes": alg[k], "ratio": round(hbm / alg[k], 3)}
    print(json.dumps(out, indent=1))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import fused
dev = torch.device("cuda", 0)
q, k, v = (torch.randn(Bw, 64, C, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
bias = (torch.randn(a.heads, 64, 64, device=dev) * 0.3).requires_grad_(True)
region = torch.randint(0, 3, (a.nw, 64), device=dev)
mask = (region[:, :, None] != region[:, None, :]).float() * -100.0
do = torch.randn(Bw, 64, C, device=dev).bfloat16()
for _ in range(a.iters):
    for t in (q, k, v, bias): t.grad = None
    fused.window_attention(q, k, v, bias, mask, a.heads, 1 / math.sqrt(dh)).backward(do)
torch.cuda.synchronize()
