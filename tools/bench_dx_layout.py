"""dX = dY @ W of the encoders' Linears: the library on W as stored ([N, K] row-major, "NN") vs on a transposed copy
([K, N], i.e. F.linear(dY, Wt), the forward's operand layout), interleaved rounds, device time by HIP events (us)."""
import json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    for M in (1024 * 197, 1024 * 77):
        for N, Kd in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):   # Linear(Kd -> N): W [N, Kd], dY [M, N], dX [M, Kd]
            dy = torch.randn(M, N, device=dev).bfloat16()
            w = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
            wt = w.t().contiguous()
            res = {"nn": [], "nt": [], "tr": []}
            for _ in range(3):
                res["nn"].append(timeit(lambda: dy @ w))
                res["nt"].append(timeit(lambda: torch.nn.functional.linear(dy, wt)))
                res["tr"].append(timeit(lambda: w.t().contiguous()))
            err = ((dy @ w).float() - torch.nn.functional.linear(dy, wt).float()).abs().max().item()
            print(json.dumps({"M": M, "N": N, "K": Kd, **{k: [round(x, 1) for x in v] for k, v in res.items()}, "max_abs_diff": err}), flush=True)


if __name__ == "__main__":
    main()
