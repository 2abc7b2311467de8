"""Segment clocks of the 8-phase weight-gradient kernel (csrc/wgrad.hip wgrad8_kernel): where a phase spends its cycles, per wave of
workgroup 0, and the in-kernel clock (shader cycles / 100 MHz real-time ticks).  Needs the stamps build:
    make -C mmlearn_amd/csrc VARIANT=_stamps EXTRA="-DMMK_DEBUG_SWITCHES -DMMK_WGRAD_STAMPS_BUILD" -j8
    MMK_LIB_VARIANT=_stamps MMK_WGRAD_STAMPS=1 python tools/wgrad_stamps.py
Read SHARES from it, not run time: the stamps fence the schedule (guide 7, "In-kernel stamps")."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import _lib, kernels as K
dev = torch.device("cuda", 0)
M = int(os.environ.get("M", 1024 * 197))
for N, K_ in ((768, 768), (3072, 768)):
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K_, device=dev).bfloat16()
    for _ in range(int(os.environ.get("WARM", 20))):   # sustained load first: the clock under load is what matters
        K.wgrad(dy, x)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    _lib.check(_lib.lib().mmk_wgrad_debug_stamps(C.cast(buf, C.c_void_p)))
    rows = []
    for w in range(8):
        mf, w2, ld, w1, tot, rt, nt, _ = (int(buf[w * 8 + i]) for i in range(8))
        ph = 4 * nt
        rows.append({"wave": w, "k_tiles": nt, "cycles_per_phase": round(tot / ph, 1), "mfma_cluster": round(mf / ph, 1), "wait_closing_barrier": round(w2 / ph, 1),
                     "load_segment": round(ld / ph, 1), "wait_opening_barrier": round(w1 / ph, 1), "clock_GHz": round(tot / rt * 0.1, 3)})
    print(json.dumps({"M": M, "N": N, "K": K_, "per_wave": rows}), flush=True)
