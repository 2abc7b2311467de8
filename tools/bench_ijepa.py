"""I-JEPA path-op microbenchmark (single GPU): HIP-event times and achieved HBM GB/s of the row kernels at the
BASELINE config-5 shape (B=128 per GPU, 196 patches, D=1024, 4 target blocks, bf16) and of the EMA update at
ViT-L size (304 M parameters).

    python tools/bench_ijepa.py
"""

import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mmlearn_amd import _lib, ops  # noqa: E402
from mmlearn_amd import kernels as K  # noqa: E402
from mmlearn_amd.masking import IJEPAMaskGenerator  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    return {k: ms / cnt * 1e3 for k, (cnt, ms) in prof.items()}  # us per launch


def main():
    dev = torch.device("cuda", 0)
    B, N, D = 128, 196, 1024
    torch.manual_seed(0)
    mi = IJEPAMaskGenerator()(batch_size=B)
    pred_idx, enc_idx = mi["predictor_indices"].to(dev), mi["encoder_indices"].to(dev)
    keep, n_ctxt = pred_idx.shape[-1], enc_idx.shape[-1]
    T = 4 * B * keep
    h = torch.randn(B, N, D, device=dev)                      # teacher output (f32 after LayerNorm under autocast)
    z = torch.randn(4 * B, keep, D, device=dev).bfloat16().requires_grad_(True)
    out = {"shape": {"B": B, "patches": N, "D": D, "keep": keep, "n_ctxt": n_ctxt, "rows_T": T}}

    def loss_step():
        z.grad = None
        ops.ijepa_loss(z, h, pred_idx).backward()

    t = timed(loss_step)
    fwd_bytes = T * D * (4 + 2)          # read h rows (f32) + z rows (bf16)
    bwd_bytes = T * D * (4 + 2 + 2)      # read h, z; write dz
    out["ijepa_loss_fwd"] = {"us": round(t["ijepa_loss_fwd"], 2), "GBps": round(fwd_bytes / t["ijepa_loss_fwd"] / 1e3, 1)}
    out["ijepa_loss_bwd"] = {"us": round(t["ijepa_loss_bwd"], 2), "GBps": round(bwd_bytes / t["ijepa_loss_bwd"] / 1e3, 1)}

    x = torch.randn(B, N, D, device=dev).bfloat16().requires_grad_(True)

    def gather_step():
        x.grad = None
        ops.gather_patches(x, enc_idx).sum().backward()

    t = timed(gather_step)
    out["gather_rows(ctx)"] = {"us": round(t["gather_rows"], 2), "GBps": round(2 * B * n_ctxt * D * 2 / t["gather_rows"] / 1e3, 1)}
    out["scatter_rows(ctx)"] = {"us": round(t["scatter_rows"], 2), "GBps": round((B * n_ctxt + B * N) * D * 2 / t["scatter_rows"] / 1e3, 1)}

    Dp = 384
    xe = torch.randn(B, n_ctxt, Dp, device=dev).bfloat16().requires_grad_(True)
    pos = torch.randn(1, N, Dp, device=dev)
    tok = torch.randn(1, 1, Dp, device=dev, requires_grad=True)

    def asm_step():
        xe.grad = None
        tok.grad = None
        ops.predictor_assemble(xe, pos, tok, enc_idx, pred_idx, B).sum().backward()

    t = timed(asm_step)
    seq_rows = 4 * B * (n_ctxt + keep)
    out["pred_assemble"] = {"us": round(t["pred_assemble"], 2), "GBps": round((seq_rows * Dp * 4 + B * n_ctxt * Dp * 2 * 4) / t["pred_assemble"] / 1e3, 1)}
    out["pred_assemble_bwd"] = {"us": round(t["pred_assemble_bwd"], 2)}

    # EMA at ViT-L size: 304 M f32 parameters in ~300 tensors
    sizes = [1024 * 1024] * 96 + [4096 * 1024] * 48 + [1024] * 200 + [4096] * 48
    student = [torch.randn(n, device=dev) for n in sizes]
    teacher = [torch.zeros(n, device=dev) for n in sizes]
    P = sum(sizes)
    tab = K.ema_table(teacher, student)
    for mode, name, bytes_per in ((False, "ema_copy", 8), (True, "ema_true", 12)):
        t = timed(lambda: K.ema_update(*tab, 0.996, mode), iters=10, warmup=2)
        out[name] = {"params_M": round(P / 1e6, 1), "us": round(t["ema_update"], 1), "GBps": round(P * bytes_per / t["ema_update"] / 1e3, 1),
                     "frac_of_8TBps": round(P * bytes_per / t["ema_update"] / 1e3 / HBM_PEAK_GBS, 3)}

    # the reference's eager sequence for the target path, for scale (torch ops on the same GPU)
    masks = [m.to(dev) for m in mi["predictor_masks"]]

    def eager_target_loss():
        hn = torch.nn.functional.layer_norm(h, (D,))
        tg = torch.cat([hn[mk.bool().unsqueeze(-1).expand(-1, -1, D)].view(B, -1, D) for mk in masks], 0)
        zz = z.detach().requires_grad_(True)
        torch.nn.functional.smooth_l1_loss(zz.float(), tg).backward()

    for _ in range(3):
        eager_target_loss()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eager_target_loss()
    torch.cuda.synchronize()
    out["torch_eager_target_loss_fwd_bwd_us"] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
