"""Attention backward: the shipped kernel (five-product, one launch) against the two-kernel form (dK/dV launch + dQ launch, images
double-buffered) -- interleaved same-process A/B in the debug-switch build (MMK_ATTN_SPLIT is read per call),
HIP-event kernel durations (both launches of the split form summed), plus parity of the split form against the shipped one.
    MMK_LIB_VARIANT=_dbg python tools/bench_attn_split.py [--out profiles/r04_attn_bwd.json]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K, _lib

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
rows = []
for B, H, L, p in ((1024, 12, 197, 0.0), (1024, 12, 77, 0.1), (512, 16, 169, 0.0)):
    q, k, v = (torch.randn(B, H, L, 64, device=dev).bfloat16() for _ in range(3))
    do = torch.randn(B, L, H, 64, device=dev).bfloat16()     # the gradient of the [B, L, H, 64] output
    scale = 0.125
    o, lse = K.attn_fwd(q, k, v, scale, p, 3)
    res = {}
    times = {"0": [], "1": []}
    for rnd in range(args.rounds):
        for sw in ("0", "1"):
            os.environ["MMK_ATTN_SPLIT"] = sw
            for _ in range(2): out = K.attn_bwd(q, k, v, o, lse, do, scale, p, 3)
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_read()
            for _ in range(8): out = K.attn_bwd(q, k, v, o, lse, do, scale, p, 3)
            torch.cuda.synchronize()
            pr = _lib.profile_read(); _lib.profile_enable(False)
            times[sw].append(pr["attn_bwd"][1] / 8 * 1e3)      # us per backward call (one or two launches)
            res[sw] = out
    err = max((a.float() - b.float()).abs().max().item() / max(1e-6, b.float().abs().max().item()) for a, b in zip(res["1"], res["0"]))
    row = {"B": B, "H": H, "L": L, "dropout": p, "shipped_us": round(statistics.median(times["0"]), 1), "two_kernel_us": round(statistics.median(times["1"]), 1),
           "shipped_min_us": round(min(times["0"]), 1), "two_kernel_min_us": round(min(times["1"]), 1), "max_rel_diff_of_gradients": err}
    print(json.dumps(row), flush=True)
    rows.append(row)
if args.out:
    json.dump({"tool": "tools/bench_attn_split.py", "note": "main kernels only (the delta pre-pass, ~90-130 us, is common to both forms)", "rows": rows}, open(args.out, "w"), indent=1)
