"""Windowed attention (csrc/window_attention.hip) at HTSAT's four resolutions, batch 256: device time of forward and backward against
their algorithmic HBM bytes (forward: q, k, v in, o out; backward: q, k, v, dO in, dq, dk, dv out; [64 x 24] bf16 each per window
and head, + lse), and the HF op sequence (two batched matmuls, bias / mask adds, f32 softmax under bf16 autocast) on the same tensors.
    python tools/bench_window_attn.py [--out gpurun_out/window_attn.json] [--batch 256]"""
import argparse, json, math, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import fused

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)


def hf_sequence(q, k, v, bias, mask, heads):
    Bw, T, C = q.shape
    dh = C // heads
    qh, kh, vh = (t.view(Bw, T, heads, dh).transpose(1, 2) for t in (q, k, v))
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(dh)
    s = s + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        s = (s.view(Bw // nW, nW, heads, T, T) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, T, T)
    p = torch.nn.functional.softmax(s, dim=-1)
    return torch.matmul(p, vh).permute(0, 2, 1, 3).contiguous().view(Bw, T, C)


def timed(fn, rounds):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 200)
    return statistics.median(ts)


rows = []
for nW, heads, masked in ((64, 4, False), (64, 4, True), (16, 8, True), (4, 16, True), (1, 32, False)):
    dh, B = 24, args.batch
    C, Bw = heads * dh, args.batch * nW
    q, k, v = (torch.randn(Bw, 64, C, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    bias = (torch.randn(heads, 64, 64, device=dev) * 0.3).requires_grad_(True)
    mask = None
    if masked:
        region = torch.randint(0, 3, (nW, 64), device=dev)
        mask = (region[:, :, None] != region[:, None, :]).float() * -100.0
    do = torch.randn(Bw, 64, C, device=dev).bfloat16()
    res = {"windows_per_sample": nW, "heads": heads, "masked": masked, "items": Bw * heads}
    for arm, fn in (("hip", lambda: fused.window_attention(q, k, v, bias, mask, heads, 1 / math.sqrt(dh))),
                    ("hf_ops", lambda: hf_sequence(q, k, v, bias, mask, heads))):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fwd = timed(lambda: fn(), args.rounds)
            def both():
                for t in (q, k, v, bias): t.grad = None
                fn().backward(do)
            fb = timed(both, args.rounds)
        res[arm + "_fwd_us"] = round(fwd, 1); res[arm + "_fwd_bwd_us"] = round(fb, 1)
    unit = Bw * 64 * C * 2
    res["fwd_algorithmic_bytes"] = 4 * unit + Bw * heads * 64 * 4
    res["bwd_algorithmic_bytes"] = 7 * unit + Bw * heads * 64 * 4
    res["hip_fwd_GBps"] = round(res["fwd_algorithmic_bytes"] / res["hip_fwd_us"] * 1e-3, 1)
    res["hip_bwd_GBps"] = round(res["bwd_algorithmic_bytes"] / max(1e-9, res["hip_fwd_bwd_us"] - res["hip_fwd_us"]) * 1e-3, 1)
    print(json.dumps(res), flush=True); rows.append(res)
if args.out:
    json.dump({"tool": "tools/bench_window_attn.py", "batch": args.batch, "note": "fwd_bwd includes the autograd glue (table build, partial sums of the bias gradient, gradient accumulation)", "rows": rows}, open(args.out, "w"), indent=1)
