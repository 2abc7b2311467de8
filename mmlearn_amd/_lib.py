"""ctypes binding of ``libmmlearn_hip.so`` (the C ABI declared in ``include/mmlearn_hip.h``).

This is the stub a maintainer of the reference would add.  There is no
CPU fallback: every op of this package goes through the HIP library, and loading fails
loudly (``RuntimeError``) when it is missing.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMK_LIB_VARIANT=_dbg: the -DMMK_DEBUG_SWITCHES build (`make -C mmlearn_amd/csrc VARIANT=_dbg EXTRA=-DMMK_DEBUG_SWITCHES`), for
# measurement tools only; the product library itself reads no experiment switches
LIB_PATH = os.path.join(_HERE, "lib", f"libmmlearn_hip{os.environ.get('MMK_LIB_VARIANT', '')}.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

F32, BF16, F16 = 0, 1, 2
COMPUTE_BF16, COMPUTE_F32 = 1, 0
ABI_VERSION = 8   # include/mmlearn_hip.h MMK_ABI_VERSION this ctypes mirror was written against

_DTYPE_TAG = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

KERNEL_NAMES = [
    "match_ids", "pack_rows", "transpose", "sim_stats", "lse_reduce", "loss_combine", "sim_grad", "grad_gemm",
    "grad_finalize", "l2norm", "ijepa_loss_fwd", "ijepa_loss_bwd", "gather_rows", "scatter_rows", "pred_assemble",
    "pred_assemble_bwd", "ema_update", "mask_to_index", "layernorm_fwd", "layernorm_bwd", "activation", "attn_fwd", "attn_bwd", "wgrad", "recall_ranks", "clip_fused", "mlp_gemm",
    "win_attn_fwd", "win_attn_bwd", "clip_bwd_fused",
]


class ClipDir(C.Structure):
    """mirror of ``mmk_clip_dir``"""

    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("yT", C.c_void_p),
        ("r", C.c_int32), ("c", C.c_int32), ("label_off", C.c_int32), ("ldt", C.c_int32),
        ("part", C.c_void_p), ("diag", C.c_void_p), ("lse", C.c_void_p), ("loss_part", C.c_void_p),
        ("lse_col", C.c_void_p), ("g", C.c_void_p), ("ldg", C.c_int32),
        ("c_row", C.c_float), ("c_col", C.c_float), ("c_diag", C.c_float),
        ("s_row", C.c_float), ("s_col", C.c_float), ("s_diag", C.c_float),
        ("kappa", C.c_float), ("ds_kappa", C.c_float),
        ("slab", C.c_void_p), ("ds_part", C.c_void_p),
        ("dx", C.c_void_p), ("dx_rows", C.c_void_p), ("dx_dtype", C.c_int32), ("dx_accumulate", C.c_int32),
        ("src", C.c_void_p), ("src_dtype", C.c_int32), ("normalize", C.c_int32),
        ("mode", C.c_int32), ("hmax", C.c_void_p),
        ("mirror_part", C.c_void_p), ("mirror_lse", C.c_void_p), ("mirror_loss_part", C.c_void_p),
        ("gT", C.c_void_p), ("ldgt", C.c_int32), ("g_ready", C.c_int32),
        ("x_norm", C.c_void_p), ("y_norm", C.c_void_p),
        ("g_transposed", C.c_int32), ("tn_ws", C.c_void_p), ("tn_ws_floats", C.c_int64),
    ]


class PackReq(C.Structure):
    """mirror of ``mmk_pack_req``"""

    _fields_ = [("src", C.c_void_p), ("idx", C.c_void_p), ("dst", C.c_void_p), ("dstT", C.c_void_p),
                ("r", C.c_int32), ("r_pad", C.c_int32), ("normalize", C.c_int32), ("ldt", C.c_int32), ("norm", C.c_void_p)]


class FusedPair(C.Structure):
    """mirror of ``mmk_fused_pair``"""

    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("idx_a", C.c_void_p), ("idx_b", C.c_void_p),
                ("n", C.c_int32), ("weight", C.c_float), ("da", C.c_void_p), ("db", C.c_void_p),
                ("da_accumulate", C.c_int32), ("db_accumulate", C.c_int32)]


class EmaEntry(C.Structure):
    """mirror of ``mmk_ema_entry``"""

    _fields_ = [("teacher", C.c_void_p), ("student", C.c_void_p), ("numel", C.c_int64),
                ("teacher_dtype", C.c_int32), ("student_dtype", C.c_int32)]


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
# name -> argtypes; every function returns int (0 = ok) unless noted
_SIGNATURES = {
    "mmk_abi_version": [],
    "mmk_device_check": [],
    "mmk_profile_enable": [_i],
    "mmk_profile_read": [_vp, _vp],
    "mmk_match_ids": [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "mmk_clip_mirror_tiles": [_i],
    "mmk_pack_rows_many": [_vp, _i, _i, _i, _i, _i, _vp],
    "mmk_clip_forward_loss": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "mmk_clip_tickets": [_vp, _i],
    "mmk_pack_rows": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mmk_clip_plan": [_i, _i, _i, _i, _vp, _vp, _vp],
    "mmk_clip_forward": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mmk_reduce_sums": [_vp, _vp, _vp, _i, _i, _vp, _vp],
    "mmk_clip_fused_plan": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "mmk_clip_fused_forward": [_vp, _i, _i, _i, _vp, _vp, C.c_int64, _i, _vp, _vp, _vp],
    "mmk_clip_fused_backward": [_vp, _i, _i, _i, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp],
    "mmk_clip_fused_debug_stamps": [_vp],
    "mmk_match_workspace_ints": [_i, _i],
    "mmk_clip_backward": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "mmk_clip_backward_plan": [_vp, _i, _i, _i, _vp],
    "mmk_l2norm_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mmk_l2norm_fwd_twin": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmk_l2norm_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmk_mask_to_index": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "mmk_gather_rows": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmk_scatter_rows": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mmk_ijepa_loss_blocks": [_i],
    "mmk_ijepa_loss_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp],
    "mmk_ijepa_loss_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "mmk_pred_tok_blocks": [_i],
    "mmk_pred_assemble": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "mmk_pred_assemble_bwd": [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "mmk_ema_update": [_vp, _i, C.c_int64, _f, _i, _vp],
    "mmk_ema_update_dev": [_vp, _i, C.c_int64, _vp, _i, _vp],
    "mmk_layernorm_part_blocks": [C.c_long],
    "mmk_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _f, _i, _vp],
    "mmk_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _vp],
    "mmk_add_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _f, _i, _f, C.c_uint64, _vp],
    "mmk_add_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _f, C.c_uint64, _vp],
    "mmk_bias_act_part_blocks": [C.c_long],
    "mmk_patchify": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mmk_unpatchify": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mmk_cast_transpose": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mmk_recall_ranks": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "mmk_adamw_chunk_elems": [],
    "mmk_adamw_update": [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, C.c_int64, _vp],
    "mmk_adamw_update_dev": [_vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _vp, _vp],
    "mmk_embedding_bwd": [_vp, _vp, _vp, C.c_int64, _i, C.c_int64, _i, _vp],
    "mmk_cubic_resize_rows": [_vp, _vp, C.c_int64, _i, _i, _i, _i, _vp],
    "mmk_colsum_rows_slices": [C.c_int64],
    "mmk_colsum_rows": [_vp, C.c_int64, _i, _i, _vp, _vp, _vp],
    "mmk_embedding_bwd_scratch_bytes": [C.c_int64, _i],
    "mmk_win_attn_supported": [_i, _i, _i],
    "mmk_win_attn_blocks": [_i, _i, _i],
    "mmk_win_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _i, _i, _i, _i, _vp],
    "mmk_win_attn_bwd": [_vp] * 10 + [_i, _i, _i, _i, _i, C.c_float, _i, _i, _i, _i, _vp],
    "mmk_embedding_bwd_sorted": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, C.c_int64, _i, _vp],
    "mmk_wgrad_plan": [C.c_int64, _i, _i, _vp, _vp],
    "mmk_wgrad_debug_stamps": [_vp],
    "mmk_wgrad_partial": [_vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp],
    "mmk_wgrad": [_vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, _i, _vp],
    "mmk_bias_act_fwd": [_vp, _vp, _vp, C.c_int64, _i, _i, _i, _vp],
    "mmk_bias_act_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _vp],
    "mmk_colsum_f32": [_vp, _i, _i, _vp, _vp, _vp],
    "mmk_mlp_gemm_supported": [C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64],
    "mmk_mlp_gemm_part_rows": [C.c_int64],
    "mmk_mlp_gemm_plain": [_vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, _vp],
    "mmk_mlp_gemm_fwd_act": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, _i, _vp],
    "mmk_mlp_gemm_bwd_dact": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _vp],
    "mmk_mlp_gemm_fwd_act_grad": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, _i, _vp],
    "mmk_mlp_gemm_bwd_mul": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp],
    "mmk_quick_gelu_fwd": [_vp, _vp, C.c_int64, _i, _vp],
    "mmk_quick_gelu_bwd": [_vp, _vp, _vp, C.c_int64, _i, _vp],
    "mmk_attn_key_bias": [_vp, _i, _i, _i, C.c_int64, _vp, _vp],
    "mmk_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _f, C.c_uint64, _vp, _i, _vp],
    "mmk_attn_bwd_has_colsum": [_i],
    "mmk_attn_debug_stamps": [_vp, _i],
    "mmk_cls_attn_supported": [_i, _i],
    "mmk_cls_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, C.c_int64, _f, _f, C.c_uint64, _vp, _vp],
    "mmk_cls_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _f, _f, C.c_uint64, _vp, _vp],
    "mmk_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, C.c_uint64, _vp, _vp, _i, _vp],
}
_STR_FUNCS = {"mmk_last_error": [], "mmk_kernel_name": [_i]}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_STR_FUNCS))

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile libmmlearn_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libmmlearn_hip.so failed")
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: mmlearn_amd has no CPU/eager fallback. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C mmlearn_amd/csrc`."
            )
        l = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_int
        l.mmk_embedding_bwd_scratch_bytes.restype = C.c_int64
        for name, args in _STR_FUNCS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_char_p
        if l.mmk_abi_version() != ABI_VERSION:   # a stale .so next to newer Python: struct layouts would not match
            raise RuntimeError(f"{LIB_PATH} has ABI version {l.mmk_abi_version()}, this package needs {ABI_VERSION}: rebuild it "
                               "(`make -C mmlearn_amd/csrc`)")
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libmmlearn_hip: {lib().mmk_last_error().decode()} (rc={rc})")


def dtype_tag(dt: torch.dtype) -> int:
    try:
        return _DTYPE_TAG[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}; supported: float32, bfloat16, float16") from None


def require_gpu(t: torch.Tensor, what: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"mmlearn_amd ops run on MI355X only; got a {t.device} {what}. There is no CPU fallback."
        )


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (fast path: no Stream object)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:  # private API moved: fall back to the public one
        return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------ profiling
def profile_enable(on: bool) -> None:
    check(lib().mmk_profile_enable(int(on)))


def profile_read() -> dict:
    n = len(KERNEL_NAMES)
    cnt = (C.c_int32 * n)()
    ms = (C.c_double * n)()
    check(lib().mmk_profile_read(C.cast(cnt, _vp), C.cast(ms, _vp)))
    return {KERNEL_NAMES[k]: (cnt[k], ms[k]) for k in range(n) if cnt[k]}
