"""Interleaved same-process A/B of two settings of one debug switch of the weight-gradient kernel on the encoder shapes, random
data, HIP-event durations of the main kernel.  Needs the debug-switch build (the product build has no switch):
    make -C mmlearn_amd/csrc VARIANT=_dbg EXTRA=-DMMK_DEBUG_SWITCHES -j8 && MMK_LIB_VARIANT=_dbg python tools/bench_wgrad_ab.py
Default: the MFMA shape (MMK_WGRAD_MFMA = 16 vs 32, round 3).  AB_VAR / AB_A / AB_B choose another switch and its two values, e.g.
    AB_VAR=MMK_WGRAD_PF AB_A=1 AB_B=0      (round 4: L2 prefetch two stages ahead on / off)
The JSON keys keep the round-3 names: "16" = setting A, "32" = setting B."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K, _lib
dev = torch.device("cuda", 0)
SHAPES = [(M, N, K_) for M in (1024 * 197, 1024 * 77) for N, K_ in ((768, 768), (2304, 768), (768, 3072), (3072, 768))]
if os.environ.get("SHAPES") == "htsat":   # HTSAT at batch 256: four resolutions, packed q|k|v, out, fc1, fc2 (tools/bench_wgrad.py)
    SHAPES = [(256 * t, n * c, k * c) for t, c in ((4096, 96), (1024, 192), (256, 384), (64, 768)) for n, k in ((3, 1), (1, 1), (4, 1), (1, 4))]
if os.environ.get("SHAPES") == "ijepa":   # the 384-wide predictor of the I-JEPA step
    SHAPES = [(M, N, K_) for M in (16384, 53760) for N, K_ in ((384, 384), (1152, 384), (1536, 384), (384, 1536))]
rounds = int(os.environ.get("ROUNDS", 6))
AB_VAR, AB = os.environ.get("AB_VAR", "MMK_WGRAD_MFMA"), {"16": os.environ.get("AB_A", "16"), "32": os.environ.get("AB_B", "32")}
for M, N, K_ in SHAPES:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K_, device=dev).bfloat16()
    ref = None
    times = {"16": [], "32": []}
    totals = {"16": [], "32": []}
    for r in range(rounds):
        for v in ("16", "32"):
            os.environ[AB_VAR] = AB[v]
            for _ in range(2): K.wgrad(dy, x)
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_read()
            for _ in range(10): out = K.wgrad(dy, x)
            torch.cuda.synchronize()
            pr = _lib.profile_read(); _lib.profile_enable(False)
            times[v].append(pr["wgrad"][1] / pr["wgrad"][0] * 1e3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # main kernel + reduction, as the step sees it
            e0.record()
            for _ in range(10): K.wgrad(dy, x)
            e1.record(); torch.cuda.synchronize()
            totals[v].append(e0.elapsed_time(e1) * 100.0)
            if ref is None:
                ref = out.clone()
            else:
                assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    fl = 2.0 * M * N * K_
    print(json.dumps({"switch": AB_VAR, "A": AB["16"], "B": AB["32"], "M": M, "N": N, "K": K_, "us_16x16x32": round(med["16"], 1), "us_32x32x16": round(med["32"], 1),
                      "min_16": round(min(times["16"]), 1), "min_32": round(min(times["32"]), 1),
                      "total_us_A": round(sorted(totals["16"])[len(totals["16"]) // 2], 1), "total_us_B": round(sorted(totals["32"])[len(totals["32"]) // 2], 1),
                      "TF_16": round(fl / med["16"] / 1e6, 1), "TF_32": round(fl / med["32"] / 1e6, 1), "ratio": round(med["32"] / med["16"], 3)}), flush=True)
