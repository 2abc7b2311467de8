"""Attention backward, what the packed-QKV operand layout and the in-kernel bias column sums cost: the same kernel on
(a) contiguous [B, H, L, 64] operands, gradients into three [B, L, H, 64] buffers, (b) q/k/v as views of one [B, L, 3, H, 64] buffer
(4.6 KB between a head's rows), gradients into three buffers, (c) = (b) with the packed gradient buffer, (d) = (c) + column sums.
Interleaved same-process rounds, HIP-event kernel durations of the main kernel.
    python tools/bench_attn_layout.py [--out gpurun_out/attn_layout.json]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K, _lib

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
rows = []
for B, H, L, p in ((1024, 12, 197, 0.0), (1024, 12, 77, 0.1)):
    qkv = torch.randn(B, L, 3, H, 64, device=dev).bfloat16()
    qp, kp, vp = (qkv[:, :, i].transpose(1, 2) for i in range(3))           # [B, H, L, 64] views, row stride 3 * H * 64
    qc, kc, vc = (t.contiguous() for t in (qp, kp, vp))
    do = torch.randn(B, L, H, 64, device=dev).bfloat16()
    scale = 0.125
    o, lse = K.attn_fwd(qc, kc, vc, scale, p, 3)
    arms = {
        "contiguous": lambda: K.attn_bwd(qc, kc, vc, o, lse, do, scale, p, 3),
        "packed_operands": lambda: K.attn_bwd(qp, kp, vp, o, lse, do, scale, p, 3),
        "packed_operands_and_gradient": lambda: K.attn_bwd(qp, kp, vp, o, lse, do, scale, p, 3, packed=True),
        "packed_with_column_sums": lambda: K.attn_bwd(qp, kp, vp, o, lse, do, scale, p, 3, packed=True, colsum=True),
    }
    times = {k: [] for k in arms}
    for rnd in range(args.rounds):
        for name, fn in arms.items():
            for _ in range(2): fn()
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_read()
            for _ in range(8): fn()
            torch.cuda.synchronize()
            pr = _lib.profile_read(); _lib.profile_enable(False)
            times[name].append(pr["attn_bwd"][1] / 8 * 1e3)
    row = {"B": B, "H": H, "L": L, "dropout": p, **{k + "_us": round(statistics.median(v), 1) for k, v in times.items()}}
    print(json.dumps(row), flush=True)
    rows.append(row)
if args.out:
    json.dump({"tool": "tools/bench_attn_layout.py", "rows": rows}, open(args.out, "w"), indent=1)
