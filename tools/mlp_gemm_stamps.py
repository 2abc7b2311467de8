"""Where a K step of csrc/mlp_gemm.hip goes: shader-clock stamps of workgroup 0 on its second tile (debug-switch build only:
`make -C mmlearn_amd/csrc VARIANT=_dbg EXTRA=-DMMK_DEBUG_SWITCHES`, then MMK_LIB_VARIANT=_dbg python tools/mlp_gemm_stamps.py).
Per step and wave: [step start, last MFMA issued (before the closing wait), after the vmcnt wait, after the barrier]."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmlearn_amd import kernels as K

dev = torch.device("cuda", 0)
M, E, H = 1024 * 197, 768, 3072
dy = torch.randn(M, E, device=dev).bfloat16()
w2t = (torch.randn(H, E, device=dev) / 55).bfloat16()
g = torch.rand(M, H, device=dev).bfloat16()
for name, fn in (("plain", lambda: K.mlp_gemm_plain(dy, w2t)), ("bwd_mul", lambda: K.mlp_gemm_bwd_mul(dy, w2t, g))):
    buf = torch.zeros(8 * 64 * 4, dtype=torch.int64, device=dev)
    for _ in range(3): fn()
    os.environ["MMK_MLP_GEMM_STAMPS"] = str(buf.data_ptr())
    fn(); torch.cuda.synchronize()
    del os.environ["MMK_MLP_GEMM_STAMPS"]
    t = buf.view(8, 64, 4).cpu()
    out = {"kernel": name, "waves": {}}
    for w in (0, 3, 4, 7):
        steps = []
        for s in range(12):
            a, b, c, d = (int(x) for x in t[w, s])
            nxt = int(t[w, s + 1, 0]) if s < 11 else None
            steps.append({"compute": b - a, "vmcnt_wait": c - b, "barrier_wait": d - c})
        out["waves"][w] = {"step_total_mean": round(sum(x["compute"] + x["vmcnt_wait"] + x["barrier_wait"] for x in steps) / 12),
                           "compute_mean": round(sum(x["compute"] for x in steps) / 12), "vmcnt_wait_mean": round(sum(x["vmcnt_wait"] for x in steps) / 12),
                           "barrier_wait_mean": round(sum(x["barrier_wait"] for x in steps) / 12),
                           "per_step_compute": [x["compute"] for x in steps], "per_step_vmcnt": [x["vmcnt_wait"] for x in steps],
                           "per_step_barrier": [x["barrier_wait"] for x in steps],
                           "tile_start_to_epilogue_end": int(t[w, 21, 0]) - int(t[w, 0, 0]), "epilogue": int(t[w, 21, 0]) - int(t[w, 11, 3]),
                           "epilogue_compute_part": int(t[w, 20, 0]) - int(t[w, 11, 3])}
    print(json.dumps(out))
