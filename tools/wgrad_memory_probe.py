"""Where does the weight-gradient kernel's time go: memory or the on-chip schedule?  The same launches (same M, N, K, same grid, same
instruction stream) with the operands' ROW STRIDE set to 0 -- every contraction row aliases row 0, so after the first touch every LDS-DMA
piece is an L1 / L2 hit -- against the real strides.  (Outputs of the stride-0 runs are meaningless; only the durations matter.)
    MMK_LIB_VARIANT=_dbg python tools/wgrad_memory_probe.py        # both kernels (MMK_WGRAD_KERNEL = 8 / 2), interleaved"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib
from mmlearn_amd._lib import check, ptr, stream, dtype_tag
dev = torch.device("cuda", 0)
M = 1024 * 197
for N, K_ in ((768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K_, device=dev).bfloat16()
    splits, wsf = C.c_int(0), C.c_int64(0)
    check(_lib.lib().mmk_wgrad_plan(M, N, K_, C.cast(C.pointer(splits), C.c_void_p), C.cast(C.pointer(wsf), C.c_void_p)))
    ws = torch.empty(wsf.value, dtype=torch.float32, device=dev)
    dw = torch.empty((N, K_), dtype=torch.float32, device=dev)
    res = {"M": M, "N": N, "K": K_, "splits": splits.value}
    for rnd in range(3):
        for kern in ("8", "2"):
            os.environ["MMK_WGRAD_KERNEL"] = kern
            for name, (ldy, ldx) in (("real", (N, K_)), ("stride0", (0, 0))):
                def run():
                    check(_lib.lib().mmk_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(ws), M, N, K_, ldy, ldx, K_, dtype_tag(torch.float32), stream()))
                for _ in range(2): run()
                torch.cuda.synchronize()
                _lib.profile_enable(True); _lib.profile_read()
                for _ in range(8): run()
                torch.cuda.synchronize()
                pr = _lib.profile_read(); _lib.profile_enable(False)
                res.setdefault(f"k{kern}_{name}_us", []).append(round(pr["wgrad"][1] / pr["wgrad"][0] * 1e3, 1))
    fl = 2.0 * M * N * K_
    for k in list(res):
        if k.endswith("_us"):
            res[k.replace("_us", "_TFs")] = round(fl / sorted(res[k])[1] / 1e6, 1)
    print(json.dumps(res), flush=True)
