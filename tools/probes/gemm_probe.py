"""ctypes wrapper of the retired round-2 GEMM experiments (tools/probes/gemm.hip, gemm4.hip): `make -C mmlearn_amd/csrc probes`
builds mmlearn_amd/lib/libmmlearn_probes.so.  NOT part of the product: nothing under mmlearn_amd/ imports this."""
import ctypes as C
import os
import sys
from typing import Optional

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mmlearn_amd._lib import check as _check_main, dtype_tag, ptr, require_gpu, stream  # noqa: E402

_PATH = os.path.join(ROOT, "mmlearn_amd", "lib", "libmmlearn_probes.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} is missing: build it with `make -C mmlearn_amd/csrc probes`")
        l = C.CDLL(_PATH)
        i, vp = C.c_int, C.c_void_p
        for name, args in {"mmk_gemm_nt_supported": [C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64],
                           "mmk_gemm_nt": [vp, vp, vp, vp, vp, C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64, i, i, vp],
                           "mmk_gemm4_nt_supported": [C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64],
                           "mmk_gemm4_nt": [vp, vp, vp, vp, vp, C.c_int64, i, i, C.c_int64, C.c_int64, C.c_int64, i, i, vp]}.items():
            getattr(l, name).argtypes = args
            getattr(l, name).restype = C.c_int
        l.mmk_last_error.restype = C.c_char_p
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libmmlearn_probes: {lib().mmk_last_error().decode()} (rc={rc})")


GEMM_ACT = {None: 0, "none": 0, "quick_gelu": 1, "gelu": 2}


def gemm_nt_supported(M: int, N: int, K: int, lda: int, ldb: int, ldc: int) -> bool:
    return bool(lib().mmk_gemm_nt_supported(M, N, K, lda, ldb, ldc))


def gemm_nt(a2: torch.Tensor, b2: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
            out_dtype: torch.dtype = torch.bfloat16, want_pre: bool = False):
    """C [M, N] = a2 [M, K] @ b2 [N, K]^T (+ bias f32[N]) (-> act): bf16 operands (rows contiguous), f32 accumulation, one
    persistent HIP kernel (csrc/gemm.hip).  ``want_pre`` (with an activation) also returns the pre-activation tensor."""
    require_gpu(a2)
    assert a2.dtype == torch.bfloat16 and b2.dtype == torch.bfloat16 and a2.dim() == 2 and b2.dim() == 2
    assert a2.stride(1) == 1 and b2.stride(1) == 1 and a2.shape[1] == b2.shape[1]
    M, K_ = a2.shape
    N = b2.shape[0]
    c = torch.empty((M, N), dtype=out_dtype, device=a2.device)
    pre = torch.empty((M, N), dtype=out_dtype, device=a2.device) if want_pre else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    check(lib().mmk_gemm_nt(ptr(a2), ptr(b2), ptr(c), ptr(pre), ptr(bias), M, N, K_, a2.stride(0), b2.stride(0), c.stride(0),
                                 dtype_tag(out_dtype), GEMM_ACT[act], stream()))
    return (c, pre) if want_pre else c


def gemm4_nt(a2: torch.Tensor, b2: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
             out_dtype: torch.dtype = torch.bfloat16, want_pre: bool = False):
    """``gemm_nt`` on the four-wave kernel (csrc/gemm4.hip: 128 x 128 register tile per wave, one barrier per K step)."""
    require_gpu(a2)
    assert a2.dtype == torch.bfloat16 and b2.dtype == torch.bfloat16 and a2.dim() == 2 and b2.dim() == 2
    assert a2.stride(1) == 1 and b2.stride(1) == 1 and a2.shape[1] == b2.shape[1]
    M, K_ = a2.shape
    N = b2.shape[0]
    c = torch.empty((M, N), dtype=out_dtype, device=a2.device)
    pre = torch.empty((M, N), dtype=out_dtype, device=a2.device) if want_pre else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    check(lib().mmk_gemm4_nt(ptr(a2), ptr(b2), ptr(c), ptr(pre), ptr(bias), M, N, K_, a2.stride(0), b2.stride(0), c.stride(0),
                                  dtype_tag(out_dtype), GEMM_ACT[act], stream()))
    return (c, pre) if want_pre else c


