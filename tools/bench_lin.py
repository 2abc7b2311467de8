"""Forward / dX GEMM shapes of the encoders: csrc/gemm.hip vs torch (hipBLASLt), device time by HIP events.
    python tools/bench_lin.py [--iters 20]"""
import argparse, json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import gemm_probe as GP  # retired GEMM experiments: `make -C mmlearn_amd/csrc probes`


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default="vit")
    ap.add_argument("--desync", default="", help="comma list of MMK_GEMM_DESYNC values")
    ap.add_argument("--dbgs", default="", help="comma list of MMK_GEMM_DBG ablation masks to time in this process (interleaved rounds)")
    ap.add_argument("--vars", default="", help="comma list of MMK_GEMM_VAR values to A/B in this process (interleaved rounds)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    Mv, Mt = 1024 * 197, 1024 * 77
    shapes = {"vit": [(Mv, 2304, 768), (Mv, 768, 768), (Mv, 3072, 768), (Mv, 768, 3072), (Mv, 768, 2304)],
              "bert": [(Mt, 2304, 768), (Mt, 768, 768), (Mt, 3072, 768), (Mt, 768, 3072)],
              "loss": [(8192, 8192, 512), (1024, 1024, 512)],
              "ijepa": [(128 * 196, 3072, 1024), (128 * 196, 1024, 1024), (128 * 196, 4096, 1024), (128 * 196, 1024, 4096)]}[args.shapes]
    for M, N, Kd in shapes:
        a = torch.randn(M, Kd, device=dev).bfloat16()
        b = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
        if args.desync:
            res = {}
            for rnd in range(3):
                for v in args.desync.split(","):
                    os.environ["MMK_GEMM_DESYNC"] = v
                    res.setdefault(v, []).append(timeit(lambda: GP.gemm_nt(a, b), args.iters))
            t_lib = timeit(lambda: torch.nn.functional.linear(a, b), args.iters)
            print(json.dumps({"M": M, "N": N, "K": Kd, "hipblaslt_us": round(t_lib, 1), **{f"desync{v}_us": [round(x, 1) for x in ts] for v, ts in res.items()}}), flush=True)
            continue
        if args.dbgs:
            res = {}
            for rnd in range(3):
                for v in args.dbgs.split(","):
                    os.environ["MMK_GEMM_DBG"] = v
                    res.setdefault(v, []).append(timeit(lambda: GP.gemm_nt(a, b), args.iters))
            os.environ["MMK_GEMM_DBG"] = "0"
            print(json.dumps({"M": M, "N": N, "K": Kd, **{f"dbg{v}_us": [round(x, 1) for x in ts] for v, ts in res.items()}}), flush=True)
            continue
        if args.vars:
            res = {}
            for rnd in range(3):
                for v in args.vars.split(","):
                    os.environ["MMK_GEMM_VAR"] = v
                    res.setdefault(v, []).append(timeit(lambda: GP.gemm_nt(a, b), args.iters))
            t_lib = timeit(lambda: torch.nn.functional.linear(a, b), args.iters)
            print(json.dumps({"M": M, "N": N, "K": Kd, "hipblaslt_us": round(t_lib, 1), **{f"var{v}_us": [round(x, 1) for x in ts] for v, ts in res.items()}}), flush=True)
            continue
        if os.environ.get("MMK_GEMM_DBG") and int(os.environ["MMK_GEMM_DBG"]) & 16:
            for _ in range(30):
                c = GP.gemm_nt(a, b)
            torch.cuda.synchronize()
            st = c.view(-1)[:8].view(torch.int64).tolist()
            if int(os.environ["MMK_GEMM_DBG"]) & 32:
                raw = c.view(-1)[:4 * 52].view(torch.int64).tolist()
                for w in range(2):
                    t = raw[2 + 25 * w: 2 + 25 * (w + 1)]
                    base = raw[2]
                    print(f"wave {4 * w}: " + " | ".join(" ".join(str(t[6 * k + i] - base) for i in range(6)) for k in range(4)) + f" | end {t[24] - base}")
            print(json.dumps({"M": M, "N": N, "K": Kd, "wg0_cycles": st[0], "wg0_realtime_ticks_100MHz": st[1], "clock_GHz": round(st[0] / st[1] * 0.1, 3),
                              "wg0_us": st[1] / 100.0}), flush=True)
            continue
        t_hip = timeit(lambda: GP.gemm_nt(a, b), args.iters)
        t_lib = timeit(lambda: torch.nn.functional.linear(a, b), args.iters)
        fl = 2.0 * M * N * Kd
        out = {"M": M, "N": N, "K": Kd, "hip_us": round(t_hip, 1), "hip_tflops": round(fl / t_hip / 1e6, 1),
               "hipblaslt_us": round(t_lib, 1), "hipblaslt_tflops": round(fl / t_lib / 1e6, 1)}
        if M % 256 == 0 and N % 256 == 0:   # the four-wave kernel (csrc/gemm4.hip); MMK_GEMM4_DBG=4: no C stores (timing only)
            t4 = timeit(lambda: GP.gemm4_nt(a, b), args.iters)
            out.update({"hip4_us": round(t4, 1), "hip4_tflops": round(fl / t4 / 1e6, 1)})
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
