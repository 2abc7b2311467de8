"""MLP fc1 forward: library GEMM + bias_act kernel vs csrc/gemm.hip with the bias + activation (+ pre-activation) epilogue;
fc2 dX: library GEMM + bias_act_bwd kernel vs csrc/gemm.hip with the act'(pre) * (.) epilogue.   Device time by HIP events."""
import json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import gemm_probe as GP  # retired GEMM experiments: `make -C mmlearn_amd/csrc probes`


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return round(st.elapsed_time(en) / iters * 1e3, 1)


def main():
    dev = torch.device("cuda", 0)
    for M, act in ((1024 * 197, "quick_gelu"), (1024 * 77, "gelu")):
        E, H = 768, 3072
        x = torch.randn(M, E, device=dev).bfloat16()
        w1 = (torch.randn(H, E, device=dev) / E ** 0.5).bfloat16()
        b1 = torch.randn(H, device=dev) * 0.1
        w2 = (torch.randn(E, H, device=dev) / H ** 0.5).bfloat16()
        dy = torch.randn(M, E, device=dev).bfloat16()
        a = K.ACT_QUICK_GELU if act == "quick_gelu" else K.ACT_GELU
        h = torch.nn.functional.linear(x, w1)
        w2t = w2.t().contiguous()    # [H, E]: dAct = dy @ w2 = linear(dy, w2t)
        out = {"M": M, "act": act}
        out["fwd_lib_gemm_us"] = timeit(lambda: torch.nn.functional.linear(x, w1))
        out["fwd_bias_act_us"] = timeit(lambda: K.bias_act_fwd(h, b1, a))
        out["fwd_fused_pre_us"] = timeit(lambda: GP.gemm_nt(x, w1, b1, act, want_pre=True))
        out["fwd_fused_nopre_us"] = timeit(lambda: GP.gemm_nt(x, w1, b1, act))
        out["fwd_own_plain_us"] = timeit(lambda: GP.gemm_nt(x, w1))
        dact = torch.nn.functional.linear(dy, w2t)
        out["bwd_lib_gemm_us"] = timeit(lambda: torch.nn.functional.linear(dy, w2t))
        out["bwd_lib_gemm_nn_us"] = timeit(lambda: dy @ w2)
        out["bwd_bias_act_us"] = timeit(lambda: K.bias_act_bwd(h, b1, dact, a))
        out["bwd_own_plain_us"] = timeit(lambda: GP.gemm_nt(dy, w2t))
        if hasattr(K, "gemm_nt_dact"):
            out["bwd_fused_us"] = timeit(lambda: GP.gemm_nt_dact(dy, w2t, h, b1, act))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
