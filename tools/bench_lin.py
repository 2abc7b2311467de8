"""Forward / dX GEMM shapes of the encoders: csrc/gemm.hip vs torch (hipBLASLt), device time by HIP events.
    python tools/bench_lin.py [--iters 20]"""
import argparse, json, os, sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K


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
        t_hip = timeit(lambda: K.gemm_nt(a, b), args.iters)
        t_lib = timeit(lambda: torch.nn.functional.linear(a, b), args.iters)
        fl = 2.0 * M * N * Kd
        print(json.dumps({"M": M, "N": N, "K": Kd, "hip_us": round(t_hip, 1), "hip_tflops": round(fl / t_hip / 1e6, 1),
                          "hipblaslt_us": round(t_lib, 1), "hipblaslt_tflops": round(fl / t_lib / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
