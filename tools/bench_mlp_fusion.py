"""MLP fc1 forward and fc2 dX: library GEMM + bias_act kernel against csrc/mlp_gemm.hip with the activation in the epilogue.
Interleaved same-process A/B (ROUNDS rounds x ITERS launches per arm, medians over the rounds), device time by HIP events.

    python tools/bench_mlp_fusion.py [--out gpurun_out/mlp_fusion.json] [--rounds 5] [--iters 10]
"""
import argparse
import json
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K


def time_arms(arms, rounds, iters):
    """arms: {name: fn}.  -> {name: {"median_us", "min_us"}} from `rounds` interleaved rounds of `iters` launches each."""
    for fn in arms.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    samples = {k: [] for k in arms}
    for _ in range(rounds):
        for name, fn in arms.items():
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(iters):
                fn()
            en.record()
            torch.cuda.synchronize()
            samples[name].append(st.elapsed_time(en) / iters * 1e3)
    return {k: {"median_us": round(statistics.median(v), 1), "min_us": round(min(v), 1)} for k, v in samples.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tuned", action="store_true", help="load the shipped TunableOp selections for the library arms")
    args = ap.parse_args()
    if args.tuned:
        from mmlearn_amd import tuned
        tuned.enable()
    dev = torch.device("cuda", 0)
    results = []
    for M, act_name in ((1024 * 197, "quick_gelu"), (1024 * 77, "gelu")):
        E, H = 768, 3072
        act = K.ACT_QUICK_GELU if act_name == "quick_gelu" else K.ACT_GELU
        x = torch.randn(M, E, device=dev).bfloat16()
        w1 = (torch.randn(H, E, device=dev) / E ** 0.5).bfloat16()
        b1 = torch.randn(H, device=dev) * 0.1
        w2 = (torch.randn(E, H, device=dev) / H ** 0.5).bfloat16()
        dy = torch.randn(M, E, device=dev).bfloat16()
        h = F.linear(x, w1)
        w2t = w2.t().contiguous()    # [H, E]: dAct = dy @ w2 = linear(dy, w2t)
        dact = F.linear(dy, w2t)
        _, gfac = K.mlp_gemm_fwd_act_grad(x, w1, b1, act)
        arms = {
            "fwd_lib_gemm": lambda: F.linear(x, w1),
            "fwd_bias_act": lambda: K.bias_act_fwd(h, b1, act),
            "fwd_own_plain": lambda: K.mlp_gemm_plain(x, w1),
            "fwd_own_fused": lambda: K.mlp_gemm_fwd_act(x, w1, b1, act),
            "bwd_lib_gemm": lambda: F.linear(dy, w2t),
            "bwd_bias_act": lambda: K.bias_act_bwd(h, b1, dact, act),
            "bwd_own_plain": lambda: K.mlp_gemm_plain(dy, w2t),
            "bwd_own_fused": lambda: K.mlp_gemm_bwd_dact(dy, w2t, h, b1, act),
            "fwd_own_fused_grad": lambda: K.mlp_gemm_fwd_act_grad(x, w1, b1, act),
            "bwd_own_mul": lambda: K.mlp_gemm_bwd_mul(dy, w2t, gfac),
        }
        t = time_arms(arms, args.rounds, args.iters)
        flop = 2.0 * M * E * H
        out = {"M": M, "act": act_name, "E": E, "H": H, "arms": t,
               "fwd_pair_us": round(t["fwd_lib_gemm"]["median_us"] + t["fwd_bias_act"]["median_us"], 1),
               "bwd_pair_us": round(t["bwd_lib_gemm"]["median_us"] + t["bwd_bias_act"]["median_us"], 1),
               "own_plain_pflops": round(flop / t["bwd_own_plain"]["median_us"] * 1e-9, 3),
               "lib_pflops": round(flop / t["bwd_lib_gemm"]["median_us"] * 1e-9, 3)}
        out["fwd_fused_over_pair"] = round(t["fwd_own_fused"]["median_us"] / out["fwd_pair_us"], 3)
        out["bwd_fused_over_pair"] = round(t["bwd_own_fused"]["median_us"] / out["bwd_pair_us"], 3)
        # what the product runs: forward that leaves act' behind + backward that multiplies by it
        out["product_fwd_bwd_us"] = round(t["fwd_own_fused_grad"]["median_us"] + t["bwd_own_mul"]["median_us"], 1)
        out["unfused_fwd_bwd_us"] = round(out["fwd_pair_us"] + out["bwd_pair_us"], 1)
        print(json.dumps(out), flush=True)
        results.append(out)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"tool": "tools/bench_mlp_fusion.py", "rounds": args.rounds, "iters": args.iters, "tuned_library": bool(args.tuned),
                       "shapes": results}, f, indent=1)


if __name__ == "__main__":
    main()
