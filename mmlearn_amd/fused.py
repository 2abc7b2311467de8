"""Encoder-side fused ops (the first "next" row): drop-in HIP replacements for the
HBM-bound modules that dominate the non-GEMM time of the encoder step under bf16 autocast.

* :class:`LayerNorm` -- ``torch.nn.LayerNorm`` subclass (same parameters / state_dict) whose forward and backward
  are one HIP kernel each (f32 statistics, fused dx + dgamma/dbeta).  Under autocast ``F.layer_norm`` returns
  f32 and the following ``Linear`` re-casts it to bf16 in a separate kernel; ``low_precision_out=True`` emits the
  autocast dtype directly -- bit-identical for consumers that are autocast ``Linear`` layers (pre-LN blocks such as
  CLIP's ``layer_norm1/2``), not for post-LN residual streams (BERT), where it must stay off.
* :class:`QuickGELU` -- HF ``QuickGELUActivation`` (``x * sigmoid(1.702 x)``: 3 ATen kernels forward, ~6 backward)
  as one kernel each way.
* :func:`fuse_qkv_attention` -- HF ``CLIPAttention`` / ``BertSelfAttention`` keep their three ``Linear`` projections
  (same parameters, same state_dict) but run them as ONE ``[E -> 3E]`` GEMM whose packed ``[B, L, 3, H, 64]`` output
  goes straight into the HIP attention kernels, and whose packed gradient comes straight out of them: 3 + 6 GEMMs with
  N = K = 768 become 1 + 2 with N or K = 2304, and the two ``dX`` accumulation kernels disappear
  (4.27 -> 2.57 ms per ViT-B/16 layer at B = 1024 for the projections alone).
* :func:`fuse_add_layer_norm` -- the blocks' ``residual + sublayer(...)`` (+ BERT's hidden-state dropout) runs inside
  the following LayerNorm's kernel, forward and backward (the backward also absorbs the gradient-accumulation add and
  the f32 -> bf16 cast of the sublayer gradient).
* :func:`patch_conv_as_gemm` -- a ``Conv2d`` whose stride equals its kernel (ViT patch embedding) runs as HIP im2col +
  one GEMM (with the HIP weight gradient) instead of MIOpen's implicit GEMM and its layout transposes.
* :func:`accelerate_encoder` swaps those modules in place inside any encoder (HF CLIP / BERT, mmlearn's own ViT).

There is no CPU path: CPU tensors raise.
"""

from __future__ import annotations

import inspect
import math
import types
from typing import Iterable, Optional

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import kernels as K


def _autocast_bf16() -> bool:
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        w32 = None if weight is None else weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        y, mean, rstd = K.layernorm_fwd(x2, w32, b32, eps, out_dtype)
        ctx.save_for_backward(x2, w32, mean, rstd)
        ctx.shape = x.shape
        ctx.wb = (weight is not None and weight.requires_grad, bias is not None and bias.requires_grad,
                  None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w32, mean, rstd = ctx.saved_tensors
        need_w, need_b, wdt, bdt = ctx.wb
        dy2 = dy.contiguous().view(x2.shape)
        dx, dw, db = K.layernorm_bwd(x2, dy2, w32, mean, rstd, need_w or need_b)
        return (dx.view(ctx.shape), dw.to(wdt) if need_w else None, db.to(bdt) if need_b else None, None, None)


def layer_norm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float = 1e-5,
               low_precision_out: bool = False) -> torch.Tensor:
    """``F.layer_norm(x, x.shape[-1:], weight, bias, eps)`` on MI355X.  Output dtype: f32 under autocast (like torch) or
    the autocast dtype when ``low_precision_out``; ``x.dtype`` outside autocast."""
    K.require_gpu(x)
    d = x.shape[-1]
    if d % 4 or d > 2048:
        raise ValueError(f"mmlearn_amd.fused.layer_norm supports a normalised dim that is a multiple of 4 and <= 2048, got {d}")
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError(f"unsupported dtype {x.dtype}")
    if torch.is_autocast_enabled():
        out_dtype = torch.get_autocast_dtype("cuda") if low_precision_out else torch.float32
    else:
        out_dtype = x.dtype
    return _LayerNormFn.apply(x, weight, bias, eps, out_dtype)


class LayerNorm(nn.LayerNorm):
    """``torch.nn.LayerNorm`` over the last dimension with HIP forward/backward kernels."""

    low_precision_out: bool = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if len(self.normalized_shape) != 1:
            raise ValueError("mmlearn_amd.fused.LayerNorm normalises the last dimension only")
        return layer_norm(x, self.weight, self.bias, self.eps, self.low_precision_out)

    @classmethod
    def from_torch(cls, ln: nn.LayerNorm, low_precision_out: bool = False) -> "LayerNorm":
        new = cls.__new__(cls)
        nn.Module.__init__(new)
        new.normalized_shape, new.eps, new.elementwise_affine = ln.normalized_shape, ln.eps, ln.elementwise_affine
        new.weight, new.bias = ln.weight, ln.bias      # the SAME Parameter objects: optimizers / checkpoints are unaffected
        new.low_precision_out = low_precision_out
        new.train(ln.training)
        return new


class _AddLayerNormFn(torch.autograd.Function):
    """(s, y[, y16]) = (r + dropout(x + xbias), LN(s)[, bf16 copy of y]) in one kernel each way (csrc/encoder_ops.hip)."""

    @staticmethod
    def forward(ctx, x, r, weight, bias, xbias, eps, out_dtype, dropout_p, seed, twin):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        r2 = r.contiguous().view(-1, d)
        w32 = None if weight is None else weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        xb32 = None if xbias is None else xbias.detach().float().contiguous()
        s, y, mean, rstd, y16 = K.add_layernorm_fwd(x2, r2, w32, b32, eps, out_dtype, dropout_p, seed, xb32, twin)
        ctx.save_for_backward(s, w32, mean, rstd)
        ctx.shape, ctx.x_dtype, ctx.dropout_p, ctx.seed = x.shape, x.dtype, dropout_p, seed
        ctx.wb = (weight is not None and weight.requires_grad, bias is not None and bias.requires_grad,
                  None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        ctx.xb = (xbias is not None and xbias.requires_grad, None if xbias is None else xbias.dtype)
        ctx.set_materialize_grads(False)  # an unused output (s in post-LN blocks) must arrive as None, not as a zeros tensor
        if twin:
            return s.view(x.shape), y.view(x.shape), y16.view(x.shape)
        return s.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, gs, gy, gy16=None):
        s, w32, mean, rstd = ctx.saved_tensors
        d = s.shape[-1]
        need_w, need_b, wdt, bdt = ctx.wb
        need_xb, xbdt = ctx.xb
        if gy is None and gy16 is None:
            gy = torch.zeros(ctx.shape, dtype=torch.float32, device=s.device)
        ds_in = None if gs is None else gs.contiguous().view(-1, d).float()
        dr, dx, dw, db, dxb = K.add_layernorm_bwd(s, None if gy is None else gy.contiguous().view(-1, d), ds_in, w32, mean, rstd, ctx.x_dtype,
                                                  need_w or need_b, ctx.dropout_p, ctx.seed, need_xb,
                                                  None if gy16 is None else gy16.contiguous().view(-1, d).to(torch.bfloat16))
        return (dx.view(ctx.shape), dr.view(ctx.shape), dw.to(wdt) if need_w else None, db.to(bdt) if need_b else None,
                dxb.to(xbdt) if need_xb else None, None, None, None, None, None)


def add_layer_norm(x: torch.Tensor, residual: torch.Tensor, ln: nn.LayerNorm, dropout_p: float = 0.0, seed: Optional[int] = None,
                   low_precision_out: Optional[bool] = None, xbias: Optional[torch.Tensor] = None, twin: bool = False):
    """``s = residual + dropout(x + xbias)``, ``y = ln(s)`` fused; returns ``(s, y)``.  ``x`` is the sublayer output
    (autocast dtype or f32), ``residual`` the f32 stream.  ``y`` is f32 like ``F.layer_norm`` under autocast unless
    ``low_precision_out`` (default: the module's own ``low_precision_out`` flag) asks for the autocast dtype.
    ``xbias``: bias of the Linear that produced ``x`` when it was run bias-free (``linear_nobias``); its gradient is
    then a by-product of this op's backward instead of a separate pass over the Linear's output gradient.
    ``twin`` (f32 ``y`` only): also emit a bf16 copy of ``y`` and attach it as ``y._mmk_bf16`` -- in post-LN blocks ``y`` is
    both the f32 residual stream and the input of the next GEMM; ``fused.linear`` picks the copy up instead of casting,
    and the two gradients meeting at this fork are summed inside the backward kernel (no cast, no add kernel)."""
    K.require_gpu(x)
    if residual.dtype != torch.float32:
        residual = residual.float()
    if low_precision_out is None:
        low_precision_out = bool(getattr(ln, "low_precision_out", False))
    out_dtype = torch.float32
    if low_precision_out and torch.is_autocast_enabled():
        out_dtype = torch.get_autocast_dtype("cuda")
    if dropout_p > 0.0 and seed is None:
        from .attention import draw_seed

        seed = draw_seed()
    twin = bool(twin and out_dtype == torch.float32 and _autocast_bf16())
    outs = _AddLayerNormFn.apply(x, residual, ln.weight, ln.bias, xbias, ln.eps, out_dtype, float(dropout_p), int(seed or 0), twin)
    if twin:
        outs[1]._mmk_bf16 = outs[2]
    return outs[0], outs[1]


def _ln_fusable(ln, x: torch.Tensor) -> bool:
    return (isinstance(ln, nn.LayerNorm) and len(ln.normalized_shape) == 1 and x.is_cuda and x.shape[-1] == ln.normalized_shape[0]
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and x.dtype in (torch.float32, torch.bfloat16, torch.float16))


class _BiasActFn(torch.autograd.Function):
    """``act(x + bias)`` for a bias-free Linear output; the backward also yields the bias gradient."""

    @staticmethod
    def forward(ctx, x, bias, act):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        b32 = bias.detach().float().contiguous()
        ctx.save_for_backward(x2, b32)
        ctx.act, ctx.shape, ctx.bdt = act, x.shape, bias.dtype
        return K.bias_act_fwd(x2, b32, act).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, b32 = ctx.saved_tensors
        dx, db = K.bias_act_bwd(x2, b32, dy.contiguous().view(x2.shape), ctx.act)
        return dx.view(ctx.shape), db.to(ctx.bdt), None


def bias_act(x: torch.Tensor, bias: torch.Tensor, act: str) -> torch.Tensor:
    """``quick_gelu(x + bias)`` / ``gelu(x + bias)`` (erf form) in one kernel; ``x`` = ``linear_nobias(...)``."""
    K.require_gpu(x)
    return _BiasActFn.apply(x, bias, {"quick_gelu": K.ACT_QUICK_GELU, "gelu": K.ACT_GELU}[act])


def _weight_operands(w: torch.Tensor):
    """-> (bf16 w for the forward, the tensor the backward keeps, twin?).  With the twin (default) one kernel casts the master
    weight and also writes its transpose [in, out]; the backward's dX = dY W then runs as ``F.linear(dY, W^T)`` -- the
    operand layout of the forward, which the library serves 5-15 % faster than ``dY @ W`` at the encoder shapes (DESIGN.md
    5.5).  ``MMK_NO_DX_TWIN=1`` keeps the plain cast and ``dY @ W`` (A/B switch)."""
    wd = w.detach()
    if (os.environ.get("MMK_NO_DX_TWIN") or not wd.is_contiguous() or wd.shape[0] % 4 or wd.shape[1] % 4
            or wd.dtype not in (torch.float32, torch.bfloat16, torch.float16)):
        w16 = wd.to(torch.bfloat16)
        return w16, w16, False
    w16, w16t = K.cast_transpose(wd)
    return w16, w16t, True


def _dx_gemm(dy2: torch.Tensor, w_bwd: torch.Tensor, twin: bool) -> torch.Tensor:
    return F.linear(dy2, w_bwd) if twin else dy2 @ w_bwd


class _LinearWgradFn(torch.autograd.Function):
    """``x @ W^T (+ b)`` in bf16 whose weight gradient ``dY^T x`` runs on ``csrc/wgrad.hip`` (split over the rows, f32 sums,
    written straight in the parameter's dtype).  At the encoder shapes (M = batch x tokens = 201,728 / 78,848 rows) that
    GEMM is where the library is weakest -- 0.75 -> 0.30 ms for the 768 x 768 projections, 1.1-1.4 -> 0.75-0.97 ms for
    the 2304- and 3072-wide ones -- because an output of a few hundred tiles with a contraction of 10^5 rows needs the
    split over M and the shared-L2 placement more than it needs a big tile."""

    @staticmethod
    def forward(ctx, x, w, b):
        k = x.shape[-1]
        x2 = x.reshape(-1, k).to(torch.bfloat16)
        w16, w_bwd, ctx.w_twin = _weight_operands(w)
        ctx.save_for_backward(x2, w_bwd)
        ctx.x_shape, ctx.x_dtype, ctx.w_dtype = x.shape, x.dtype, w.dtype
        ctx.b_dtype = None if b is None else b.dtype
        with torch.autocast("cuda", enabled=False):
            y = x2 @ w16.t() if b is None else torch.addmm(b.detach().to(torch.bfloat16), x2, w16.t())
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w_bwd = ctx.saved_tensors
        dy2 = dy.reshape(-1, w_bwd.shape[1] if ctx.w_twin else w_bwd.shape[0]).to(torch.bfloat16).contiguous()
        dx = dw = db = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                dx = _dx_gemm(dy2, w_bwd, ctx.w_twin).view(ctx.x_shape).to(ctx.x_dtype)
            if ctx.needs_input_grad[1]:
                dw = K.wgrad(dy2, x2, ctx.w_dtype if ctx.w_dtype in (torch.float32, torch.bfloat16) else torch.float32).to(ctx.w_dtype)
            if ctx.b_dtype is not None and ctx.needs_input_grad[2]:
                db = (K.colsum_rows(dy2) if dy2.shape[1] % 8 == 0 else dy2.sum(0, dtype=torch.float32)).to(ctx.b_dtype)
        return dx, dw, db


class _QKVAttentionFn(torch.autograd.Function):
    """Packed QKV projection + attention as ONE autograd node: ``softmax(q k^T scale) v`` of ``x @ W^T + b`` split into
    heads of 64.  Same kernels as ``linear`` -> ``attention_qkvpacked``; what the fusion buys is in the backward -- the
    attention kernel hands over the column sums of the packed gradient it has just written, so the projection's bias
    gradient needs no ``dY.sum(0)`` pass over the [rows, 3E] gradient (0.19 ms per ViT-B/16 layer at B = 1024)."""

    @staticmethod
    def forward(ctx, x, w, b, heads, scale, dropout_p, seed, key_bias=None, causal=False):
        B, L, E = x.shape
        x2 = x.reshape(-1, E).to(torch.bfloat16)
        w16, w_bwd, w_twin = _weight_operands(w)
        with torch.autocast("cuda", enabled=False):
            qkv = x2 @ w16.t() if b is None else torch.addmm(b.detach().to(torch.bfloat16), x2, w16.t())
        qkv = qkv.view(B, L, 3, heads, 64)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        out, lse = K.attn_fwd(q, k, v, scale, dropout_p, seed, key_bias, causal)
        ctx.save_for_backward(x2, w_bwd, qkv, out, lse, key_bias)
        ctx.meta = (x.shape, x.dtype, w.dtype, None if b is None else b.dtype, scale, dropout_p, seed, w_twin, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, w_bwd, qkv, out, lse, key_bias = ctx.saved_tensors
        x_shape, x_dtype, w_dtype, b_dtype, scale, dropout_p, seed, w_twin, causal = ctx.meta
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        want_db = b_dtype is not None and ctx.needs_input_grad[2]
        res = K.attn_bwd(q, k, v, out, lse, dout.contiguous(), scale, dropout_p, seed, packed=True, colsum=want_db, key_bias=key_bias,
                         causal=causal)
        dqkv, db = res if want_db else (res, None)
        dy2 = dqkv.view(x2.shape[0], -1)
        dx = dw = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                dx = _dx_gemm(dy2, w_bwd, w_twin).view(x_shape).to(x_dtype)
            if ctx.needs_input_grad[1]:
                dw = K.wgrad(dy2, x2, w_dtype if w_dtype in (torch.float32, torch.bfloat16) else torch.float32).to(w_dtype)
        return dx, dw, (None if db is None else db.to(b_dtype)), None, None, None, None, None, None


def qkv_attention(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], heads: int, scale: float, dropout_p: float,
                  key_bias: Optional[torch.Tensor] = None, causal: bool = False):
    """``[B, L, E]`` -> ``[B, L, heads, 64]`` through the packed projection ``w [3E, E]`` / ``b [3E]`` and the HIP
    attention kernels; one autograd node where the weight-gradient kernel applies, two (``linear`` +
    ``attention_qkvpacked``) otherwise.  None when the projection does not come out in bf16.  ``key_bias`` / ``causal``: the
    key-padding records of ``attention.key_bias_of`` and the causal triangle (``attention.attention``)."""
    from .attention import _causal_bias, attention_qkvpacked, draw_seed

    x16 = getattr(x, "_mmk_bf16", None)   # bf16 twin attached by add_layer_norm(twin=True)
    if x16 is not None and x16.shape == x.shape and _autocast_bf16():
        x = x16
    B, L, E = x.shape
    if _wgrad_linear_ok(w, x) and L <= 256 and w.shape[0] == 3 * heads * 64 and not os.environ.get("MMK_NO_QKV_NODE"):   # (A/B switch)
        seed = draw_seed() if dropout_p > 0.0 else 0
        return _QKVAttentionFn.apply(x, w, b, int(heads), float(scale), float(dropout_p), int(seed), _causal_bias(key_bias, causal, x, B, L),
                                     bool(causal))
    qkv = linear(x, w, b)
    if qkv.dtype != torch.bfloat16:
        return None
    return attention_qkvpacked(qkv.view(B, L, 3, heads, 64), scale, dropout_p, key_bias=key_bias, causal=causal)


def _wgrad_linear_ok(weight: torch.Tensor, x: torch.Tensor) -> bool:
    rows = x.numel() // max(x.shape[-1], 1)
    return (x.is_cuda and weight.dim() == 2 and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0 and rows >= 6144
            and torch.is_grad_enabled() and weight.requires_grad
            and (x.dtype == torch.bfloat16 or _autocast_bf16()))


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.linear`` with the HIP weight-gradient kernel in its backward where that applies (bf16, >= 6k rows)."""
    x16 = getattr(x, "_mmk_bf16", None)   # bf16 twin attached by add_layer_norm(twin=True)
    if x16 is not None and x16.shape == x.shape and _autocast_bf16():
        x = x16
    if _wgrad_linear_ok(weight, x):
        return _LinearWgradFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def _linear_module_forward(self, x):
    return linear(x, self.weight, self.bias)


def _plain_linear(m) -> bool:
    """Does ``m(x)`` compute exactly ``F.linear(x, m.weight, m.bias)``?  Every fusion below reads ``.weight`` / ``.bias`` directly and
    never calls the module, which is only sound for a stock ``nn.Linear``: an adapter-wrapped projection (peft's LoRA ``Linear`` keeps
    the base layer's ``.weight`` visible while its ``forward`` adds the adapter delta), a subclass, an instance-level ``forward`` that is
    not this package's own, or a module with forward / backward hooks would silently lose its extra terms and their gradients."""
    if type(m) is not nn.Linear:
        return False
    own = m.__dict__.get("forward")
    if own is not None and getattr(own, "__func__", None) is not _linear_module_forward:
        return False
    return not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)


def linear_wgrad(module: nn.Module) -> int:
    """Patch every plain ``nn.Linear`` inside ``module`` to run through :func:`linear`: where the layer sees >= 6k rows in bf16
    its weight gradient comes from ``csrc/wgrad.hip`` (split over the rows), its dX from the transposed twin.  For towers none of
    the block-level fusions recognise -- HTSAT's Swin stages: ``dW [384 x 96] = dY^T x`` over 1,048,576 rows is 12 output tiles
    for the library's 64 x 64 x 256 kernel, 0.96 ms a call (profiles/r03_htsat_tower_kernel_stats.csv); 29 such GEMMs per step."""
    n = 0
    for m in module.modules():
        if type(m) is nn.Linear and "forward" not in m.__dict__:
            m.forward = types.MethodType(_linear_module_forward, m)
            n += 1
    return n


def linear_nobias(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """``x @ W^T`` of an ``nn.Linear`` whose bias is added by the consumer kernel (``add_layer_norm`` / ``bias_act``)."""
    return linear(x, lin.weight, None)


MLP_PAD_MIN_ROWS = 2048   # below this an MLP is too small for the persistent GEMM to pay; rows are never padded then


def _pad_rows(m: int) -> int:
    """Row count the MLP GEMMs run on: ``m`` rounded up to a multiple of 256 (unchanged when it is one, or when the MLP is small)."""
    return m if (m % 256 == 0 or m < MLP_PAD_MIN_ROWS) else (m + 255) // 256 * 256


class _MLPFn(torch.autograd.Function):
    """``act(x W1^T + b1) W2^T`` -- a two-layer MLP without fc2's bias (the consumer kernel adds it) as ONE autograd node around
    the two GEMMs of ``csrc/mlp_gemm.hip`` that carry the activation pass in their epilogue:

    * forward: ``fc1`` + bias + activation in one kernel that writes the activation and, in place of the pre-activation, the
      activation's derivative ``G = act'(x W1^T + b1)`` (the unfused step runs library GEMM -> ``bias_act_fwd``: one read of the
      [rows, hidden] tensor more);
    * backward: ``dPre = (dY W2) * G`` and ``db1 = dPre.sum(0)`` inside fc2's dX GEMM (the unfused step writes ``dY W2``, reads it
      back with the pre-activation and writes ``dPre``: two passes over [rows, hidden] more).

    fc2's forward and fc1's dX stay library GEMMs, both weight gradients run on ``csrc/wgrad.hip``.  Row counts that are not a
    multiple of 256 are padded with zero rows for the two fused kernels (``_pad_rows``); shapes the kernels do not serve at all
    (widths that are not multiples of 256 / 64, small MLPs) take the library GEMM + ``bias_act`` kernels inside the same node."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, act):
        k = x.shape[-1]
        x2 = x.reshape(-1, k).to(torch.bfloat16)
        if x2.stride(1) != 1 or x2.stride(0) % 8:
            x2 = x2.contiguous()
        w1_16, w1_bwd, ctx.w1_twin = _weight_operands(w1)
        w2_16, w2_bwd, ctx.w2_twin = _weight_operands(w2)
        b32 = b1.detach().float().contiguous()
        M, H, E = x2.shape[0], w1.shape[0], w2.shape[0]
        # The kernels walk whole 256-row tiles.  A row count that is not a multiple of 256 (I-JEPA: batch x kept patches) is padded
        # with zero rows -- one small copy of x here and of dY in the backward against a pass over [rows, hidden] each way; the
        # padded rows of act' meet zero rows of dY in the backward, so nothing of them reaches a gradient.
        Mp = _pad_rows(M)
        # both kernels or neither: the forward leaves act'(pre + b1) behind INSTEAD of the pre-activation, which only the fused
        # backward can use
        ctx.fused = bool(not os.environ.get("MMK_NO_MLP_FUSION")   # (A/B switch)
                         and ctx.w2_twin and K.mlp_gemm_supported(Mp, H, k, k if Mp != M else x2.stride(0), w1_16.stride(0), H)
                         and K.mlp_gemm_supported(Mp, H, E, E, w2_bwd.stride(0), H))
        with torch.autocast("cuda", enabled=False):
            if ctx.fused:
                xp = x2 if Mp == M else F.pad(x2, (0, 0, 0, Mp - M))
                a2, h2 = K.mlp_gemm_fwd_act_grad(xp, w1_16, b32, act)     # h2 = act'(x W1^T + b1), [Mp, H]
                a2 = a2[:M]
            else:
                h2 = x2 @ w1_16.t()                                       # h2 = x W1^T
                a2 = K.bias_act_fwd(h2, b32, act)
            y = a2 @ w2_16.t()
        ctx.save_for_backward(x2, h2, a2, b32, w1_bwd, w2_bwd)
        ctx.meta = (x.shape, x.dtype, w1.dtype, b1.dtype, w2.dtype, act)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h2, a2, b32, w1_bwd, w2_bwd = ctx.saved_tensors
        x_shape, x_dtype, w1_dtype, b_dtype, w2_dtype, act = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).contiguous()
        wdt = lambda t: t if t in (torch.float32, torch.bfloat16) else torch.float32
        dx = dw1 = db1 = dw2 = None
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[3]:
                dw2 = K.wgrad(dy2, a2, wdt(w2_dtype)).to(w2_dtype)
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                if ctx.fused:
                    M = dy2.shape[0]
                    dyp = dy2 if h2.shape[0] == M else F.pad(dy2, (0, 0, 0, h2.shape[0] - M))
                    dpre, db1 = K.mlp_gemm_bwd_mul(dyp, w2_bwd, h2, want_dbias=ctx.needs_input_grad[2])
                    dpre = dpre[:M]
                else:
                    dpre, db1 = K.bias_act_bwd(h2, b32, _dx_gemm(dy2, w2_bwd, ctx.w2_twin), act)
                if ctx.needs_input_grad[0]:
                    dx = _dx_gemm(dpre, w1_bwd, ctx.w1_twin).view(x_shape).to(x_dtype)
                if ctx.needs_input_grad[1]:
                    dw1 = K.wgrad(dpre, x2, wdt(w1_dtype)).to(w1_dtype)
                db1 = db1.to(b_dtype) if (db1 is not None and ctx.needs_input_grad[2]) else None
        return dx, dw1, db1, dw2, None


def mlp_fc1_act_fc2(x: torch.Tensor, fc1: nn.Linear, act: str, fc2: nn.Linear) -> torch.Tensor:
    """``fc2.weight`` applied to ``act(fc1(x))`` -- everything of a two-layer MLP except fc2's bias, which the consumer kernel adds.
    Where the weight-gradient path applies (bf16, >= 6k rows, gradients on) the whole thing is one autograd node on the fused
    GEMMs (``_MLPFn``); otherwise the three separate ops: bias-free fc1, bias + activation kernel, bias-free fc2."""
    x16 = getattr(x, "_mmk_bf16", None)   # bf16 twin attached by add_layer_norm(twin=True)
    if x16 is not None and x16.shape == x.shape and _autocast_bf16():
        x = x16
    if (_wgrad_linear_ok(fc1.weight, x) and fc2.weight.requires_grad and fc1.bias is not None and fc2.weight.dim() == 2
            and fc2.weight.shape[1] == fc1.weight.shape[0] and fc2.weight.shape[0] % 8 == 0 and fc2.weight.shape[1] % 8 == 0):
        return _MLPFn.apply(x, fc1.weight, fc1.bias, fc2.weight, {"quick_gelu": K.ACT_QUICK_GELU, "gelu": K.ACT_GELU}[act])
    if (not torch.is_grad_enabled() and x.is_cuda and _autocast_bf16() and fc1.bias is not None and fc1.weight.dim() == 2
            and not os.environ.get("MMK_NO_MLP_FUSION")):
        # forward only (an EMA teacher, evaluation): fc1 + bias + activation in one kernel, no second output
        k = x.shape[-1]
        x2 = x.reshape(-1, k).to(torch.bfloat16)
        M, H = x2.shape[0], fc1.weight.shape[0]
        Mp = _pad_rows(M)
        w16 = fc1.weight.detach().to(torch.bfloat16)
        if x2.stride(1) == 1 and K.mlp_gemm_supported(Mp, H, k, k if Mp != M else x2.stride(0), w16.stride(0), H):
            with torch.autocast("cuda", enabled=False):
                xp = x2 if Mp == M else F.pad(x2, (0, 0, 0, Mp - M))
                a2, _ = K.mlp_gemm_fwd_act(xp, w16, fc1.bias.detach().float().contiguous(), {"quick_gelu": K.ACT_QUICK_GELU, "gelu": K.ACT_GELU}[act],
                                           want_pre=False)
            return linear_nobias(fc2, a2[:M].view(*x.shape[:-1], H))
    return linear_nobias(fc2, bias_act(linear_nobias(fc1, x), fc1.bias, act))


def _act_name(fn) -> Optional[str]:
    name = type(fn).__name__
    if name in ("QuickGELUActivation", "QuickGELU"):
        return "quick_gelu"
    if name == "GELUActivation" and getattr(fn, "act", None) is F.gelu:
        return "gelu"
    return None


def _bias_deferrable(lin: nn.Linear, x: torch.Tensor) -> bool:
    return (_plain_linear(lin) and lin.bias is not None and x.is_cuda and lin.out_features % 8 == 0
            and (torch.is_autocast_enabled() or lin.weight.dtype == x.dtype))


def _clip_mlp_nobias(mlp, x2: torch.Tensor):
    """HF ``CLIPMLP`` as fc1 (bias-free GEMM) -> bias + activation kernel -> fc2 (bias-free GEMM); returns the bias-free
    output and fc2's bias for the consumer, or ``(mlp(x2), None)`` when that form does not apply."""
    act = _act_name(mlp.activation_fn)
    if act is None or not (_bias_deferrable(mlp.fc1, x2) and _bias_deferrable(mlp.fc2, x2)):
        return mlp(x2), None
    return mlp_fc1_act_fc2(x2, mlp.fc1, act, mlp.fc2), mlp.fc2.bias


def _clip_layer_forward(self, hidden_states, attention_mask=None, **kwargs):
    """Replaces HF ``CLIPEncoderLayer.forward``.  ``residual + attn`` runs inside ``layer_norm2``'s kernel, and
    ``residual + mlp`` inside the NEXT layer's ``layer_norm1`` kernel: this layer still returns the sum (the hidden state
    HF records), with the already-normalised tensor attached to it for the next layer to pick up.  The biases of
    ``out_proj`` / ``fc1`` / ``fc2`` are applied by the consuming kernels (their gradients fall out of those kernels'
    backward passes instead of three ``dY.sum(0)`` reductions)."""
    residual = hidden_states
    x = getattr(hidden_states, "_mmk_prenormed", None)
    if x is None:
        x = self.layer_norm1(hidden_states)
    attn = self.self_attn
    defer = hasattr(attn, "_mmk_stock_forward") and _ln_fusable(self.layer_norm2, x) and _bias_deferrable(attn.out_proj, x)
    if defer:
        attn._mmk_defer_out_bias = True
    try:
        a, _ = attn(hidden_states=x, attention_mask=attention_mask, **kwargs)
        deferred = defer and getattr(attn, "_mmk_out_bias_deferred", False)
    finally:
        if defer:
            attn._mmk_defer_out_bias = False
            attn._mmk_out_bias_deferred = False
    if _ln_fusable(self.layer_norm2, a):
        h, x2 = add_layer_norm(a, residual, self.layer_norm2, xbias=attn.out_proj.bias if deferred else None)
    else:
        h = residual + a
        x2 = self.layer_norm2(h)
    nxt = getattr(self, "_mmk_next_ln", None)
    if nxt is not None and _ln_fusable(nxt, x2):
        m, mb = _clip_mlp_nobias(self.mlp, x2)
        out, y = add_layer_norm(m, h, nxt, xbias=mb)
        out._mmk_prenormed = y
        return out
    m, mb = _clip_mlp_nobias(self.mlp, x2)   # last layer: no next LayerNorm to fuse the add into
    return h + m if mb is None else h + (m + mb)


def _bert_output_forward(self, hidden_states, input_tensor):
    """Replaces HF ``BertSelfOutput.forward`` / ``BertOutput.forward``: dense bias + dropout + residual add + LayerNorm in
    one kernel (the dense GEMM runs bias-free; the bias gradient comes out of the fused backward)."""
    if _bias_deferrable(self.dense, hidden_states):
        h = linear_nobias(self.dense, hidden_states)
        if _ln_fusable(self.LayerNorm, h):
            return add_layer_norm(h, input_tensor, self.LayerNorm, self.dropout.p if self.training else 0.0, xbias=self.dense.bias,
                                  twin=True)[1]
        h = h + self.dense.bias
    else:
        h = self.dense(hidden_states)
        if _ln_fusable(self.LayerNorm, h):
            return add_layer_norm(h, input_tensor, self.LayerNorm, self.dropout.p if self.training else 0.0)[1]
    return self.LayerNorm(self.dropout(h) + input_tensor)


def _bert_intermediate_forward(self, hidden_states):
    """Replaces HF ``BertIntermediate.forward``: bias-free dense GEMM + one bias + GELU kernel."""
    act = _act_name(self.intermediate_act_fn)
    if act is not None and _bias_deferrable(self.dense, hidden_states):
        return bias_act(linear_nobias(self.dense, hidden_states), self.dense.bias, act)
    return self.intermediate_act_fn(self.dense(hidden_states))


def _bert_ffn_chunk(self, attention_output):
    """Replaces HF ``BertLayer.feed_forward_chunk`` (``self.output(self.intermediate(x), x)``): intermediate and output are
    separate modules there, here the GELU kernel and ``output.dense`` share one autograd node (``mlp_fc1_act_fc2``: the GELU's
    backward runs inside that GEMM's dX kernel) and ``output.dense.bias`` + dropout + residual + LayerNorm stay one kernel."""
    inter, outp = self.intermediate, self.output
    act = _act_name(inter.intermediate_act_fn)
    if (act is None or not _bias_deferrable(inter.dense, attention_output) or not _bias_deferrable(outp.dense, attention_output)
            or not _ln_fusable(outp.LayerNorm, attention_output)):
        return outp(inter(attention_output), attention_output)
    y = mlp_fc1_act_fc2(attention_output, inter.dense, act, outp.dense)
    return add_layer_norm(y, attention_output, outp.LayerNorm, outp.dropout.p if self.training else 0.0, xbias=outp.dense.bias, twin=True)[1]


def _is_preln_block(m: nn.Module) -> bool:
    """Duck type of the timm-style pre-LN block mmlearn's own ViT / I-JEPA predictor are built from
    (mmlearn/modules/layers/transformer_block.py: ``norm1, attn(qkv, proj, attn_drop, proj_drop, num_heads, scale),
    drop_path, norm2, mlp``)."""
    a = getattr(m, "attn", None)
    return (all(hasattr(m, k) for k in ("norm1", "norm2", "mlp", "drop_path")) and a is not None
            and isinstance(getattr(a, "qkv", None), nn.Linear) and isinstance(getattr(a, "proj", None), nn.Linear)
            and all(hasattr(a, k) for k in ("num_heads", "scale", "attn_drop", "proj_drop")))


def _drop_p(mod, training: bool) -> Optional[float]:
    """Dropout probability of an ``nn.Dropout`` / ``nn.Identity`` (None: something this path does not understand)."""
    if isinstance(mod, nn.Identity):
        return 0.0
    if isinstance(mod, nn.Dropout):
        return float(mod.p) if training else 0.0
    return None


def _seq_mlp_nobias(mlp, x2: torch.Tensor, training: bool):
    """``Sequential(Linear, act, Dropout, Linear, Dropout)`` (mmlearn/modules/layers/mlp.py with one hidden layer) as
    fc1 (bias-free GEMM) -> bias + activation kernel -> fc2 (bias-free GEMM).  Returns (output without fc2's bias, that
    bias, dropout p after fc2) or None when the container is not of that form / has an active inner dropout."""
    if not isinstance(mlp, nn.Sequential) or len(mlp) != 5:
        return None
    fc1, act, d1, fc2, d2 = mlp
    name = "gelu" if (type(act) is nn.GELU and getattr(act, "approximate", "none") == "none") else _act_name(act)
    p1, p2 = _drop_p(d1, training), _drop_p(d2, training)
    if not (isinstance(fc1, nn.Linear) and isinstance(fc2, nn.Linear) and name is not None and p1 == 0.0 and p2 is not None
            and _bias_deferrable(fc1, x2) and _bias_deferrable(fc2, x2)):
        return None
    return mlp_fc1_act_fc2(x2, fc1, name, fc2), fc2.bias, p2


def _preln_block_forward(self, x, return_attention: bool = False):
    """Replaces the forward of a timm-style pre-LN block (mmlearn/modules/layers/transformer_block.py:125-133): fused QKV
    attention on the packed projection, ``x + proj_drop(proj(.))`` inside ``norm2``'s kernel, bias + activation kernel in
    the MLP and ``x + mlp(.)`` inside the next block's ``norm1`` kernel.  Stochastic depth, attention-map requests, head
    sizes other than 64, sequences over 256 tokens and non-bf16 runs take the stock forward."""
    from .attention import attention_qkvpacked

    attn = self.attn
    p_attn, p_proj = _drop_p(attn.attn_drop, self.training), _drop_p(attn.proj_drop, self.training)
    ok = (not return_attention and isinstance(self.drop_path, nn.Identity) and p_attn is not None and p_proj is not None
          and x.dim() == 3 and x.is_cuda and _autocast_bf16() and x.shape[1] <= 256 and x.shape[2] == 64 * attn.num_heads
          and _plain_linear(attn.qkv) and attn.qkv.out_features == 3 * x.shape[2] and _ln_fusable(self.norm2, x)
          and _bias_deferrable(attn.proj, x))
    if not ok:
        if hasattr(x, "_mmk_prenormed"):
            del x._mmk_prenormed
        return self._mmk_stock_layer_forward(x, return_attention)
    B, L, E = x.shape
    xn = getattr(x, "_mmk_prenormed", None)
    if xn is None:
        xn = self.norm1(x)
    ctx = qkv_attention(xn, attn.qkv.weight, attn.qkv.bias, attn.num_heads, float(attn.scale), p_attn)
    if ctx is None:
        return self._mmk_stock_layer_forward(x, return_attention)
    y = linear_nobias(attn.proj, ctx.reshape(B, L, E))
    h, x2 = add_layer_norm(y, x, self.norm2, p_proj, xbias=attn.proj.bias)
    fused = _seq_mlp_nobias(self.mlp, x2, self.training)
    nxt = getattr(self, "_mmk_next_ln", None)
    if fused is not None and nxt is not None and _ln_fusable(nxt, x2):
        m, mb, p2 = fused
        out, yn = add_layer_norm(m, h, nxt, p2, xbias=mb)
        out._mmk_prenormed = yn
        return out
    if fused is not None and fused[2] == 0.0:
        return h + (fused[0] + fused[1])
    return h + self.mlp(x2)


_ADD_LN_FORWARDS = {"CLIPEncoderLayer": _clip_layer_forward, "BertSelfOutput": _bert_output_forward, "BertOutput": _bert_output_forward,
                    "BertIntermediate": _bert_intermediate_forward}


def fuse_add_layer_norm(module: nn.Module) -> int:
    """Patch HF ``CLIPEncoderLayer`` / ``BertSelfOutput`` / ``BertOutput`` / ``BertIntermediate`` and timm-style pre-LN
    blocks (mmlearn's own ViT and I-JEPA predictor: recognised by their attributes, see ``_is_preln_block``) inside
    ``module`` -- in place, parameters and state_dict untouched -- so that each residual add (+ hidden dropout) runs
    inside the following LayerNorm's kernel and the Linear biases are applied by the consuming kernels."""
    n = 0
    for m in module.modules():
        fwd = _ADD_LN_FORWARDS.get(type(m).__name__)
        if fwd is None and _is_preln_block(m):
            fwd = _preln_block_forward
        if fwd is not None and not hasattr(m, "_mmk_stock_layer_forward"):
            m._mmk_stock_layer_forward = m.forward
            m.forward = types.MethodType(fwd, m)
            n += 1
        if (type(m).__name__ == "BertLayer" and "feed_forward_chunk" not in m.__dict__ and hasattr(m, "intermediate")
                and hasattr(m, "output") and not getattr(m, "add_cross_attention", False)):
            m.feed_forward_chunk = types.MethodType(_bert_ffn_chunk, m)
        if isinstance(m, nn.ModuleList) and len(m) > 1:   # consecutive pre-LN layers: see _clip_layer_forward
            if all(type(c).__name__ == "CLIPEncoderLayer" for c in m):
                for cur, nxt in zip(list(m)[:-1], list(m)[1:]):
                    object.__setattr__(cur, "_mmk_next_ln", nxt.layer_norm1)
            elif all(_is_preln_block(c) for c in m):
                for cur, nxt in zip(list(m)[:-1], list(m)[1:]):
                    object.__setattr__(cur, "_mmk_next_ln", nxt.norm1)
    return n


class _QuickGELUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return K.quick_gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.quick_gelu_bwd(x, dy.contiguous().to(x.dtype))


class QuickGELU(nn.Module):
    """``x * sigmoid(1.702 * x)`` (HF ``QuickGELUActivation``) as one HIP kernel per direction."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        K.require_gpu(x)
        if x.numel() % 4:
            raise ValueError("QuickGELU kernel needs a multiple of 4 elements")
        return _QuickGELUFn.apply(x)


class _MaskScope:
    """What a patched HF text model knows about the ``attention_mask`` of the forward in flight.  HF turns the tokenizer's 2-D mask
    (``[B, L]`` ones / zeros: mmlearn/modules/encoders/text.py:160-165, clip.py:104-107, 329-346 always forward it) into a materialised
    ``[B, 1, L, L]`` tensor per call, after a host-synchronising "is it all ones?" test, and hands THAT to every attention module -- from
    which nobody can tell any more that it masks keys only.  The model-level pre-hook below sees the 2-D mask itself: it builds the key
    bias records once per forward (``kernels.attn_key_bias``), keeps them here for the model's attention modules, and removes the mask
    from HF's view, so that no 4-D tensor is built and nothing waits for the device.  An attention module that cannot use the records
    (head size, sequence length, dtype) rebuilds an additive ``[B, 1, 1, L]`` mask from ``mask2d`` for its stock forward.  A model that
    re-runs its layers in the backward (gradient checkpointing), when this forward's records are gone, is left alone: HF's own mask
    reaches the layers in both runs and the stock forward serves it."""

    __slots__ = ("key_bias", "mask2d", "shape", "stripped")

    def __init__(self):
        self.clear()

    def clear(self):
        self.key_bias, self.mask2d, self.shape, self.stripped = None, None, None, False

    def additive(self, dtype: torch.dtype) -> torch.Tensor:
        keep = self.mask2d != 0
        return torch.zeros(keep.shape, dtype=dtype, device=keep.device).masked_fill_(~keep, torch.finfo(dtype).min)[:, None, None, :]


def _mask_argument(model, args, kwargs):
    """(where, value) of ``attention_mask`` in a call of ``model.forward``."""
    if "attention_mask" in kwargs:
        return "kw", kwargs["attention_mask"]
    names = getattr(model, "_mmk_forward_params", None)
    if names is None:
        names = [n for n in inspect.signature(type(model).forward).parameters if n != "self"]
        object.__setattr__(model, "_mmk_forward_params", names)
    if "attention_mask" in names and names.index("attention_mask") < len(args):
        return names.index("attention_mask"), args[names.index("attention_mask")]
    return None, None


def _mask_scope_pre_hook(model, args, kwargs):
    scope = getattr(model, "_mmk_mask_scope", None)
    if scope is None:
        return None
    scope.clear()
    where, mask = _mask_argument(model, args, kwargs)
    cfg = getattr(model, "config", None)
    if (not isinstance(mask, torch.Tensor) or mask.dim() != 2 or not mask.is_cuda or mask.shape[1] > 256 or mask.is_floating_point()
            and mask.dtype != torch.float32 or getattr(cfg, "is_decoder", False) or kwargs.get("encoder_hidden_states") is not None
            or kwargs.get("past_key_values") is not None or os.environ.get("MMK_NO_ATTN_MASK")):
        return None
    if model.training and any(getattr(m, "gradient_checkpointing", False) for m in model.modules()):
        return None   # the layers run again in the backward, after this forward's records are gone: both runs must see HF's own mask
    from .attention import key_bias_of

    kb = key_bias_of(mask, mask.shape[0], mask.shape[1], additive=False)
    if kb is None:
        return None
    scope.key_bias, scope.mask2d, scope.shape, scope.stripped = kb, mask, tuple(mask.shape), True
    if where == "kw":
        return args, dict(kwargs, attention_mask=None)
    return tuple(None if i == where else a for i, a in enumerate(args)), kwargs


def _mask_scope_post_hook(model, args, output):
    scope = getattr(model, "_mmk_mask_scope", None)
    if scope is not None:
        scope.clear()
    return None


_MASK_SCOPE_MODELS = ("BertModel", "CLIPTextTransformer", "CLIPTextModel")   # (CLIPTextModel IS the transformer in transformers 5)


def scope_key_masks(module: nn.Module) -> int:
    """Give every HF ``BertModel`` / CLIP text transformer inside ``module`` a :class:`_MaskScope` shared with its (patched) attention
    modules and the two hooks that fill and clear it -- the INNERMOST such model where they nest (transformers 4: ``CLIPTextModel`` wraps
    ``CLIPTextTransformer`` and hands the mask on).  Returns the number of models scoped."""
    n = 0
    for m in module.modules():
        if type(m).__name__ not in _MASK_SCOPE_MODELS or getattr(m, "_mmk_mask_scope", None) is not None:
            continue
        if any(c is not m and type(c).__name__ in _MASK_SCOPE_MODELS for c in m.modules()):
            continue
        attns = [a for a in m.modules() if type(a).__name__ in _QKV_FORWARDS and hasattr(a, "_mmk_stock_forward")]
        everyone = [a for a in m.modules() if type(a).__name__ in _QKV_FORWARDS]
        if not attns or len(attns) != len(everyone):
            continue   # an attention module that would never look at the scope: the mask stays HF's business
        scope = _MaskScope()
        object.__setattr__(m, "_mmk_mask_scope", scope)
        for a in attns:
            object.__setattr__(a, "_mmk_mask_scope", scope)
        m.register_forward_pre_hook(_mask_scope_pre_hook, with_kwargs=True)
        m.register_forward_hook(_mask_scope_post_hook)
        n += 1
    return n


def _scoped_key_bias(self, hidden_states, attention_mask):
    """-> (servable, key bias records or None, scope or None) for a call of a patched attention module."""
    scope = getattr(self, "_mmk_mask_scope", None)
    live = scope is not None and scope.key_bias is not None and scope.shape == tuple(hidden_states.shape[:2])
    if attention_mask is None:
        return True, (scope.key_bias if (live and scope.stripped) else None), (scope if live else None)
    B, L = hidden_states.shape[:2]
    from .attention import key_bias_of

    kb = key_bias_of(attention_mask, B, L) if L <= 256 else None   # provable from shape and strides alone ([B, 1, 1, L], expanded views)
    return kb is not None, kb, (scope if live else None)


def _stock_mask(self, hidden_states, attention_mask, scope):
    """The mask a stock forward must be given: the caller's, or -- when the model-level hook removed it -- an additive rebuild."""
    if attention_mask is None and scope is not None and scope.stripped:
        return scope.additive(hidden_states.dtype if hidden_states.is_floating_point() else torch.float32)
    return attention_mask


def _fused_qkv(self, hidden_states: torch.Tensor, names, scale: float, dropout_p: float, key_bias=None, causal: bool = False):
    """One GEMM for the three projections + the packed attention kernel; None when the call is not servable."""
    from .attention import attention_qkvpacked

    q, k, v = (getattr(self, n) for n in names)
    if hidden_states.dim() != 3 or not hidden_states.is_cuda or not (_plain_linear(q) and _plain_linear(k) and _plain_linear(v)):
        return None
    B, L, E = hidden_states.shape
    if E % 64 or L > 256 or q.weight.shape != (E, E) or not (0.0 <= dropout_p < 1.0):
        return None
    if not _autocast_bf16() and q.weight.dtype != torch.bfloat16:
        return None
    w = torch.cat([q.weight, k.weight, v.weight], 0)
    b = None if q.bias is None else torch.cat([q.bias, k.bias, v.bias], 0)
    out = qkv_attention(hidden_states, w, b, E // 64, scale, dropout_p, key_bias, causal)
    return None if out is None else out.reshape(B, L, E)


def _clip_attention_forward(self, hidden_states, attention_mask=None, **kwargs):
    """Replaces HF ``CLIPAttention.forward`` (modeling_clip.py): same parameters, same return contract.  Key-padding masks the model
    -level scope knows (or that are provable from the tensor: :func:`attention.key_mask_view`) run on the masked kernels; the vision
    tower passes none.  A causal call -- HF's CLIP text tower: ``is_causal=True`` with no mask, or a materialised causal + padding mask
    -- takes the masked kernels' causal triangle only in the first form (a 4-D mask of unknown content is never guessed at)."""
    causal = bool(kwargs.get("is_causal", False)) or bool(getattr(self, "is_causal", False))
    ok, kb, scope = _scoped_key_bias(self, hidden_states, attention_mask)
    if causal and attention_mask is not None:
        ok = False   # causal + a tensor mask: the tensor holds the triangle as well; only the stock forward knows how to read it
    if ok and getattr(self, "head_dim", 0) == 64:
        ctx = _fused_qkv(self, hidden_states, ("q_proj", "k_proj", "v_proj"), self.scale, self.dropout if self.training else 0.0, kb, causal)
        if ctx is not None:
            if getattr(self, "_mmk_defer_out_bias", False):   # the enclosing patched layer adds out_proj.bias itself
                self._mmk_out_bias_deferred = True
                return linear_nobias(self.out_proj, ctx), None
            return self.out_proj(ctx), None
    return self._mmk_stock_forward(hidden_states, _stock_mask(self, hidden_states, attention_mask, scope), **kwargs)


def _bert_self_attention_forward(self, hidden_states, attention_mask=None, past_key_values=None, **kwargs):
    """Replaces HF ``BertSelfAttention.forward`` (modeling_bert.py); the output ``dense`` lives in ``BertSelfOutput``.
    Key-padding masks (the tokenizer's, through the model-level scope) run on the masked kernels.  Decoder / causal configurations
    take the stock forward."""
    causal = bool(kwargs.get("is_causal", False)) or bool(getattr(self, "is_causal", False)) or bool(getattr(self, "is_decoder", False))
    ok, kb, scope = _scoped_key_bias(self, hidden_states, attention_mask)
    if ok and past_key_values is None and not causal and getattr(self, "attention_head_size", 0) == 64:
        ctx = _fused_qkv(self, hidden_states, ("query", "key", "value"), self.scaling, self.dropout.p if self.training else 0.0, kb)
        if ctx is not None:
            return ctx, None
    return self._mmk_stock_forward(hidden_states, _stock_mask(self, hidden_states, attention_mask, scope), past_key_values, **kwargs)


_QKV_FORWARDS = {"CLIPAttention": _clip_attention_forward, "BertSelfAttention": _bert_self_attention_forward}


def fuse_qkv_attention(module: nn.Module) -> int:
    """Give every HF ``CLIPAttention`` / ``BertSelfAttention`` inside ``module`` the fused-QKV forward (in place; the
    modules, their parameters and the state_dict stay as they are), and every HF text model that owns such modules the key-mask scope
    (:func:`scope_key_masks`: the tokenizer's ``attention_mask`` then runs on the masked kernels).  Calls the fused path cannot serve
    (a mask that is not a key-padding mask, KV cache, head size != 64, L > 256, no bf16 autocast) run the stock forward.  Returns the
    number of modules patched."""
    n = 0
    for m in module.modules():
        fwd = _QKV_FORWARDS.get(type(m).__name__)
        if fwd is not None and not hasattr(m, "_mmk_stock_forward"):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(fwd, m)
            n += 1
    scope_key_masks(module)
    return n


class _PatchifyFn(torch.autograd.Function):
    """``kernels.patchify`` for an image that requires grad (HTSAT's patch embedding sits behind a BatchNorm): the backward is the
    inverse permutation."""

    @staticmethod
    def forward(ctx, x, patch):
        ctx.shape, ctx.patch, ctx.dtype = tuple(x.shape), patch, x.dtype
        return K.patchify(x, patch)

    @staticmethod
    def backward(ctx, dcols):
        return K.unpatchify(dcols, ctx.shape, ctx.patch, ctx.dtype), None


def _patch_conv_forward(self, x):
    """Replaces the forward of an ``nn.Conv2d`` whose stride equals its kernel (ViT patch embedding): HIP im2col
    (a permutation + cast) and one GEMM; the result is returned as a ``[B, E, gh, gw]`` view of the ``[B, gh gw, E]``
    GEMM output, so the usual ``.flatten(2).transpose(1, 2)`` that follows is free."""
    P = self.kernel_size[0]
    bf16 = x.dtype == torch.bfloat16 or _autocast_bf16()
    if (x.dim() == 4 and x.is_cuda and bf16 and P % 4 == 0 and x.shape[2] % P == 0 and x.shape[3] % P == 0
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16)):
        B, _, H, W = x.shape
        cols = _PatchifyFn.apply(x.contiguous(), P) if (x.requires_grad and torch.is_grad_enabled()) else K.patchify(x, P)
        y = linear(cols, self.weight.view(self.out_channels, -1), self.bias)
        return y.view(B, H // P, W // P, self.out_channels).permute(0, 3, 1, 2)
    return self._mmk_stock_forward(x)


def patch_conv_as_gemm(module: nn.Module) -> int:
    """Patch every ``nn.Conv2d`` with ``stride == kernel_size`` (square, no padding / dilation / groups) inside ``module``."""
    n = 0
    for m in module.modules():
        if (type(m) is nn.Conv2d and m.kernel_size == m.stride and m.kernel_size[0] == m.kernel_size[1] and m.kernel_size[0] > 1
                and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"
                and not hasattr(m, "_mmk_stock_forward")):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(_patch_conv_forward, m)
            n += 1
    return n


class _EmbeddingFn(torch.autograd.Function):
    """``F.embedding`` whose weight gradient is the run-summing atomic scatter of ``csrc/encoder_ops.hip``."""

    @staticmethod
    def forward(ctx, ids, weight, padding_idx):
        ctx.save_for_backward(ids)
        ctx.vocab, ctx.w_dtype, ctx.padding_idx = weight.shape[0], weight.dtype, padding_idx
        return F.embedding(ids, weight, padding_idx)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        flat = ids.reshape(-1)
        if ctx.padding_idx is not None:   # rows looked up at padding_idx get no gradient: the kernel ignores ids outside the table
            flat = flat.masked_fill(flat == ctx.padding_idx, -1)
        dw = K.embedding_bwd(dout.reshape(-1, dout.shape[-1]), flat, ctx.vocab)
        return None, dw.to(ctx.w_dtype), None


def _embedding_forward(self, ids):
    if (ids.is_cuda and ids.dtype == torch.int64 and self.weight.requires_grad and torch.is_grad_enabled() and ids.numel() >= 4096
            and self.weight.dtype in (torch.float32, torch.bfloat16) and self.embedding_dim % 4 == 0):
        return _EmbeddingFn.apply(ids, self.weight, self.padding_idx)
    return self._mmk_stock_forward(ids)


def patch_embedding_backward(module: nn.Module) -> int:
    """Give ``nn.Embedding`` tables (no max_norm / sparse / scale_grad_by_freq; ``padding_idx`` is honoured: HF BERT's word table
    has one) the HIP backward."""
    n = 0
    for m in module.modules():
        if (type(m) is nn.Embedding and m.max_norm is None and not m.sparse and not m.scale_grad_by_freq
                and not hasattr(m, "_mmk_stock_forward")):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(_embedding_forward, m)
            n += 1
    return n


class _ClsAttnFn(torch.autograd.Function):
    """softmax(scale q k^T) v for ONE query per (sample, head), keys / values as one packed ``[B, L, 2, H, 64]`` tensor
    (csrc/cls_attention.hip: one wave per (sample, head), every key / value row read once each way, every gradient row written once)."""

    @staticmethod
    def forward(ctx, q, kv, scale, dropout_p, seed, key_bias=None):
        o, lse2 = K.cls_attn_fwd(q, kv, scale, dropout_p, seed, key_bias)
        ctx.save_for_backward(q, kv, lse2, key_bias)
        ctx.cfg = (scale, dropout_p, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, lse2, key_bias = ctx.saved_tensors
        dq, dkv = K.cls_attn_bwd(q, kv, do.contiguous(), lse2, *ctx.cfg, key_bias=key_bias)
        return dq, dkv, None, None, None, None


def _cls_query_attention(x: torch.Tensor, q_lin: nn.Linear, k_lin: nn.Linear, v_lin: nn.Linear, heads: int, scale: float,
                         dropout_p: float, key_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention output of query token 0 alone, ``[B, 1, E]``: keys and values of ALL tokens through one packed ``[2E, E]``
    projection (the HIP weight-gradient path of :func:`linear`), the query of token 0 only.  On the GPU in bf16 with 64-wide heads and
    L <= 256 the attention itself is the single-query kernel pair (``_ClsAttnFn``; the library's SDPA takes 1.6 ms for what moves in
    0.4); anything else (the CPU tests, other head sizes) runs ``F.scaled_dot_product_attention``."""
    B, L, E = x.shape
    dh = E // heads
    wkv = torch.cat([k_lin.weight, v_lin.weight], 0)
    bkv = torch.cat([k_lin.bias, v_lin.bias], 0) if (k_lin.bias is not None and v_lin.bias is not None) else None
    kv = linear(x, wkv, bkv).view(B, L, 2, heads, dh)                   # one stack in the backward, no zero-filled halves
    q = F.linear(x[:, :1], q_lin.weight, q_lin.bias).view(B, heads, dh).to(kv.dtype)
    if (kv.is_cuda and kv.dtype == torch.bfloat16 and dh == 64 and L <= 256 and kv.is_contiguous() and not os.environ.get("MMK_NO_CLS_ATTN")
            and getattr(K, "cls_attn_supported", None) is not None):
        from .attention import draw_seed

        seed = draw_seed() if dropout_p > 0.0 else 0
        return _ClsAttnFn.apply(q.contiguous(), kv, float(scale), float(dropout_p), seed, key_bias).view(B, 1, E)
    k, v = kv.unbind(2)
    # (key bias records are base-2 logits: back to an additive [B, 1, 1, L] mask in natural-log units for the library call)
    am = None if key_bias is None else (key_bias[:, :L] * 0.6931471805599453).clamp_(min=torch.finfo(q.dtype).min).to(q.dtype)[:, None, None, :]
    a = F.scaled_dot_product_attention(q.unsqueeze(2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=am, dropout_p=dropout_p, scale=scale)
    return a.transpose(1, 2).reshape(B, 1, E)


def _cls_forward_applies(attn, hidden_states, attention_mask, kwargs, projections=()):
    """-> (applies, key bias records or None) for the token-0 form of a last layer."""
    if not all(_plain_linear(p) for p in projections):   # e.g. a LoRA-wrapped q_proj / v_proj: the full layer calls the modules
        return False, None
    causal = bool(kwargs.get("is_causal", False)) or bool(getattr(attn, "is_causal", False)) or bool(getattr(attn, "is_decoder", False))
    if (causal or hidden_states.dim() != 3 or hidden_states.shape[1] <= 1 or kwargs.get("past_key_values") is not None
            or kwargs.get("output_attentions", False)):
        return False, None
    ok, kb, _ = _scoped_key_bias(attn, hidden_states, attention_mask)   # a key-padding mask is served, any other mask is not
    return ok, kb


def _clip_last_layer_cls_forward(self, hidden_states, attention_mask=None, **kwargs):
    """LAST ``CLIPEncoderLayer`` when only token 0 of its output is consumed (``last_hidden_state[:, 0]`` -> ``post_layernorm``
    -> projection: mmlearn/modules/encoders/clip.py:463-470, HF ``CLIPVisionTransformer``): LayerNorm 1 and the key / value
    projections run over all tokens, everything after them -- the query, the attention row, ``out_proj``, LayerNorm 2, the MLP and
    both residual adds -- for token 0 only.  Returns ``[B, 1, E]``: token 0 of what the full layer returns, and the same gradients
    for every parameter (the other tokens' outputs of the last layer reach nothing).  Masked / causal calls run the full layer."""
    attn = self.self_attn
    applies, kb = _cls_forward_applies(attn, hidden_states, attention_mask, kwargs, (attn.q_proj, attn.k_proj, attn.v_proj))
    if not applies:
        return self._mmk_full_forward(hidden_states, attention_mask, **kwargs)
    x = getattr(hidden_states, "_mmk_prenormed", None)
    if x is None:
        x = self.layer_norm1(hidden_states)
    a = _cls_query_attention(x, attn.q_proj, attn.k_proj, attn.v_proj, attn.num_heads, attn.scale, attn.dropout if self.training else 0.0, kb)
    h = hidden_states[:, :1] + attn.out_proj(a)
    return h + self.mlp(self.layer_norm2(h))


def _bert_last_layer_cls_forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                                 past_key_values=None, **kwargs):
    """LAST HF ``BertLayer`` when only the [CLS] position of its output is consumed (``last_hidden_state[:, 0]``): keys and
    values over all tokens, the rest of the layer for position 0.  Returns ``[B, 1, E]``.  Decoder / cross-attention / masked
    calls run the full layer."""
    sa = self.attention.self
    applies, kb = False, None
    if encoder_hidden_states is None and past_key_values is None and not getattr(self, "is_decoder", False):
        applies, kb = _cls_forward_applies(sa, hidden_states, attention_mask, kwargs, (sa.query, sa.key, sa.value))
    if not applies:
        return self._mmk_full_forward(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask,
                                      past_key_values=past_key_values, **kwargs)
    a = _cls_query_attention(hidden_states, sa.query, sa.key, sa.value, sa.num_attention_heads, sa.scaling,
                             sa.dropout.p if self.training else 0.0, kb)
    ao = self.attention.output(a, hidden_states[:, :1])
    return self.output(self.intermediate(ao), ao)


_CLS_LAST_LAYER = {"CLIPEncoder": ("layers", "CLIPEncoderLayer", _clip_last_layer_cls_forward),
                   "BertEncoder": ("layer", "BertLayer", _bert_last_layer_cls_forward)}


def reads_only_token0(module: nn.Module) -> tuple:
    """Is it PROVABLE that the consumer of ``module``'s HF ``CLIPEncoder`` / ``BertEncoder`` reads nothing but token 0 of the final
    hidden state?  -> ``(verdict, why)`` with ``verdict`` True (proven), False (contradicted: something reads other positions) or
    None (unknown consumer: not provable either way).  The test is on the code that consumes the encoder output, which is the tower wrapper's
    ``forward``; so it is answered only for wrappers whose ``forward`` is known:

    * a wrapper that declares it: class or instance attribute ``mmk_reads_only_token0`` (True / False) -- the tower's author states
      that ``forward`` reads token 0 only (``.image_embeds`` / ``.pooler_output`` / ``last_hidden_state[:, 0]``);
    * mmlearn's ``HFCLIPVisionEncoderWithProjection`` (mmlearn/modules/encoders/clip.py:444-470): its ``forward`` calls
      ``vision_model.encoder(...)`` and reads ``last_hidden_state[:, 0, :]`` unless ``use_all_token_embeddings`` is set;
    * a bare HF ``CLIPVisionModelWithProjection`` is NOT enough: its output object also carries ``last_hidden_state`` for the
      caller to read.

    In every case the HF configs inside must not ask for ``output_hidden_states`` / ``output_attentions`` (the per-layer tuple would
    show the shortened last entry).  Anything else -- mmlearn's ``HFTextEncoder`` (hands the whole last hidden state to a pooling
    layer, text.py:170-175), token-level heads, mean pooling -- is not provable and is refused."""
    for m in module.modules():
        cfg = getattr(m, "config", None)
        if cfg is not None and (getattr(cfg, "output_hidden_states", False) or getattr(cfg, "output_attentions", False)):
            return False, f"{type(m).__name__}.config asks for output_hidden_states / output_attentions"
    declared = getattr(module, "mmk_reads_only_token0", None)
    if declared is not None:
        return bool(declared), f"{type(module).__name__}.mmk_reads_only_token0 = {bool(declared)}"
    if type(module).__name__ == "HFCLIPVisionEncoderWithProjection" and hasattr(module, "use_all_token_embeddings"):
        if module.use_all_token_embeddings:
            return False, "use_all_token_embeddings=True reads every token"
        if getattr(module, "patch_dropout", None) is not None and getattr(module.patch_dropout, "exclude_first_token", True) is False:
            return False, "patch dropout may drop the class token"
        return True, "HFCLIPVisionEncoderWithProjection pools last_hidden_state[:, 0] (clip.py:463-470)"
    return None, f"{type(module).__name__}: forward is not a known token-0 consumer (declare mmk_reads_only_token0 = True on the tower to opt in)"


def cls_only_last_layer(module: nn.Module) -> int:
    """Opt-in, for encoders whose consumer reads ONLY token 0 of the final hidden state (CLS pooling: mmlearn's
    ``HFCLIPVisionEncoderWithProjection`` with ``use_all_token_embeddings=False``, a BERT text encoder pooled at [CLS]): the last
    layer of every HF ``CLIPEncoder`` / ``BertEncoder`` inside ``module`` computes its output for token 0 only and the encoder's
    ``last_hidden_state`` becomes ``[B, 1, E]``.  Loss and every parameter gradient are those of the full layer; what is
    dropped is work whose result nobody reads (10/12 of the last layer's GEMMs at ViT-B/16 / BERT-base).  Do NOT use it when
    anything reads other positions of the last hidden state (token-level heads, ``use_all_token_embeddings``, mean pooling).
    Returns the number of layers patched."""
    n = 0
    for m in module.modules():
        spec = _CLS_LAST_LAYER.get(type(m).__name__)
        if spec is None:
            continue
        layers = getattr(m, spec[0], None)
        if not layers:
            continue
        last = layers[len(layers) - 1]
        if type(last).__name__ != spec[1] or hasattr(last, "_mmk_full_forward"):
            continue
        last._mmk_full_forward = last.forward
        last.forward = types.MethodType(spec[2], last)
        n += 1
    return n


class _WindowAttentionFn(torch.autograd.Function):
    """softmax(scale q k^T + bias[h] + mask[w]) v per 64-token window and head, one kernel each way (csrc/window_attention.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, mask, heads, scale, grid, shift):
        n_win = (grid[0] // 8) * (grid[1] // 8) if grid is not None else (1 if mask is None else mask.shape[0])
        table = bias.detach().float()[None] if mask is None else bias.detach().float()[None] + mask.float()[:, None]
        table = table.contiguous()
        o, lse2 = K.win_attn_fwd(q, k, v, table, heads, n_win, scale, grid, shift)
        ctx.save_for_backward(q, k, v, lse2, table)
        ctx.cfg, ctx.bias_dtype = (heads, n_win, scale, grid, shift), bias.dtype
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, lse2, table = ctx.saved_tensors
        heads, n_win, scale, grid, shift = ctx.cfg
        dq, dk, dv, dtab = K.win_attn_bwd(q, k, v, do.contiguous(), lse2, table, heads, n_win, scale, grid, shift)
        return dq, dk, dv, dtab.to(ctx.bias_dtype), None, None, None, None, None


class _WindowAttentionPackedFn(torch.autograd.Function):
    """The same node on ONE packed ``[..., 3 C]`` projection output (q | k | v): the kernels read the thirds in place and write the
    three gradients into the thirds of one ``[..., 3 C]`` buffer, so the projection is one GEMM each way (and one weight-gradient
    launch, one bias column sum) instead of three, and the three input gradients need no adding up."""

    @staticmethod
    def forward(ctx, qkv, bias, mask, heads, scale, grid, shift):
        c = qkv.shape[-1] // 3
        n_win = (grid[0] // 8) * (grid[1] // 8) if grid is not None else (1 if mask is None else mask.shape[0])
        table = (bias.detach().float()[None] if mask is None else bias.detach().float()[None] + mask.float()[:, None]).contiguous()
        o, lse2 = K.win_attn_fwd(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], table, heads, n_win, scale, grid, shift)
        ctx.save_for_backward(qkv, lse2, table)
        ctx.cfg, ctx.bias_dtype = (heads, n_win, scale, grid, shift), bias.dtype
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, lse2, table = ctx.saved_tensors
        heads, n_win, scale, grid, shift = ctx.cfg
        c = qkv.shape[-1] // 3
        dq, _, _, dtab = K.win_attn_bwd(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], do.contiguous(), lse2, table, heads, n_win, scale, grid, shift)
        return dq._base, dtab.to(ctx.bias_dtype), None, None, None, None, None


def window_attention_packed(qkv: torch.Tensor, bias: torch.Tensor, mask: Optional[torch.Tensor], heads: int, scale: float,
                            grid: Optional[tuple] = None, shift: int = 0) -> torch.Tensor:
    """:func:`window_attention` on a packed bf16 ``[..., 3 C]`` tensor holding q | k | v along the last dimension; returns ``[..., C]``."""
    return _WindowAttentionPackedFn.apply(qkv.contiguous(), bias, mask, heads, scale, grid, int(shift))


def window_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: torch.Tensor, mask: Optional[torch.Tensor], heads: int,
                     scale: float, grid: Optional[tuple] = None, shift: int = 0) -> torch.Tensor:
    """``q, k, v``: bf16 ``[B * n_win, 64, heads * dh]`` (dh 24 or 32); ``bias``: ``[heads, 64, 64]`` relative-position bias (gets a
    gradient); ``mask``: ``[n_win, 64, 64]`` additive shifted-window mask or None.  Returns the context ``[B * n_win, 64, heads * dh]``.
    With ``grid = (h, w)``: ``q, k, v`` and the result are ``[B, h * w, C]`` token maps and the kernel gathers the 8 x 8 windows of the
    map rolled by ``-shift`` itself (and puts the context rows back where they came from)."""
    return _WindowAttentionFn.apply(q.contiguous(), k.contiguous(), v.contiguous(), bias, mask, heads, scale, grid, int(shift))


def _window_self_attention_forward(self, hidden_states, attention_mask=None, output_attentions=False):
    """Replaces HF ``ClapAudioSelfAttention.forward`` / ``SwinSelfAttention.forward`` (same code: per-window attention with a
    relative-position bias table and an optional shifted-window mask) when the fused kernel applies: bf16 projections, 64-token
    windows, head dim 24 / 32, no attention-probability dropout in effect, no attention maps asked for."""
    n_tok, ch = hidden_states.shape[1], hidden_states.shape[2]
    p_drop = float(getattr(self.dropout, "p", 0.0)) if self.training else 0.0
    ok = (hidden_states.is_cuda and not output_attentions and p_drop == 0.0 and ch == self.all_head_size
          and K.win_attn_supported(n_tok, self.attention_head_size, ch) and _autocast_bf16()
          and (attention_mask is None or (attention_mask.dim() == 3 and hidden_states.shape[0] % attention_mask.shape[0] == 0)))
    if not ok:
        return self._mmk_stock_forward(hidden_states, attention_mask, output_attentions)
    q, k, v = self.query(hidden_states), self.key(hidden_states), self.value(hidden_states)
    if q.dtype != torch.bfloat16:
        return self._mmk_stock_forward(hidden_states, attention_mask, output_attentions)
    bias = self.relative_position_bias_table[self.relative_position_index.view(-1)]
    bias = bias.view(n_tok, n_tok, -1).permute(2, 0, 1).contiguous()
    ctx_layer = window_attention(q, k, v, bias, attention_mask, self.num_attention_heads, 1.0 / math.sqrt(self.attention_head_size))
    self._mmk_rows = q.shape[0] * q.shape[1]
    return (ctx_layer,)


def _swin_layer_forward(self, hidden_states, input_dimensions, output_attentions=False, always_partition=False):
    """Replaces HF ``ClapAudioLayer.forward`` (Swin block: LN -> roll -> window_partition -> attention -> window_reverse -> roll back ->
    residual -> LN -> MLP -> residual) when the fused windowed attention applies and the map needs no padding: the q / k / v
    projections are per token, so they run on the un-rolled, un-partitioned map, and the kernel does the roll and the window
    partition -- and their inverses on the way out -- as address arithmetic.  Four copies of the map per layer less, each way."""
    att = self.attention.self
    height, width = input_dimensions
    if not always_partition:
        self.set_shift_and_window_size(input_dimensions)
    ws, shift = int(self.window_size), int(self.shift_size)
    n_tok = ws * ws
    p_drop = float(getattr(att.dropout, "p", 0.0)) if self.training else 0.0
    ok = (hasattr(att, "_mmk_stock_forward") and not output_attentions and ws == 8 and height % 8 == 0 and width % 8 == 0 and 0 <= shift < 8
          and hidden_states.is_cuda and p_drop == 0.0 and K.win_attn_supported(n_tok, att.attention_head_size, att.all_head_size)
          and _autocast_bf16() and hidden_states.shape[1] == height * width)
    if not ok:
        return self._mmk_stock_forward(hidden_states, input_dimensions, output_attentions, True)   # shift / window size are set already
    shortcut = hidden_states
    x = self.layernorm_before(hidden_states)
    # the three projections as ONE GEMM on the concatenated weights (a copy of three small matrices per step; autograd hands the
    # thirds of the packed weight gradient back to the three parameters)
    biases = (att.query.bias, att.key.bias, att.value.bias)
    if any(b is None for b in biases) != all(b is None for b in biases):
        return self._mmk_stock_forward(hidden_states, input_dimensions, output_attentions, True)
    w = torch.cat([att.query.weight, att.key.weight, att.value.weight], 0)
    qkv = linear(x, w, None if biases[0] is None else torch.cat(biases, 0))
    if qkv.dtype != torch.bfloat16:
        return self._mmk_stock_forward(hidden_states, input_dimensions, output_attentions, True)
    bias = att.relative_position_bias_table[att.relative_position_index.view(-1)]
    bias = bias.view(n_tok, n_tok, -1).permute(2, 0, 1).contiguous()
    mask = self.get_attn_mask(height, width, dtype=torch.float32, device=hidden_states.device)
    ctx_map = window_attention_packed(qkv, bias, mask, att.num_attention_heads, 1.0 / math.sqrt(att.attention_head_size), (height, width), shift)
    att._mmk_rows = qkv.shape[0] * qkv.shape[1]   # (host-side bookkeeping for bench.py's byte count)
    attention_output = self.attention.output(ctx_map, x)
    hidden_states = shortcut + self.drop_path(attention_output)
    layer_output = self.layernorm_after(hidden_states)
    inter, outp = self.intermediate, self.output
    act = _act_name(getattr(inter, "intermediate_act_fn", None))
    fc1, fc2 = getattr(inter, "dense", None), getattr(outp, "dense", None)
    if (act is not None and type(fc1) is nn.Linear and type(fc2) is nn.Linear and fc1.bias is not None
            and fc2.in_features == fc1.out_features):
        # the block's MLP as the encoders' MLP node (``mlp_fc1_act_fc2``: bias + GELU in the GEMM epilogue where the shape allows, one
        # bias + activation kernel each way otherwise) instead of dense -> ATen GELU -> dense: no new kernel, the same ones CLIP / BERT run
        y = mlp_fc1_act_fc2(layer_output, fc1, act, fc2)
        if fc2.bias is not None:
            y = y + fc2.bias.to(y.dtype)
        layer_output = hidden_states + outp.dropout(y)
    else:
        layer_output = hidden_states + outp(inter(layer_output))
    return (layer_output,)


def fuse_window_attention(module: nn.Module) -> int:
    """Patch every HF windowed self-attention module (``ClapAudioSelfAttention``, ``SwinSelfAttention``, ...: recognised by their
    attributes) inside ``module`` with :func:`_window_self_attention_forward`, and every Swin-style layer around one
    (``ClapAudioLayer``: ``layernorm_before`` / ``attention.self`` / ``get_attn_mask`` / ``shift_size``) with
    :func:`_swin_layer_forward`.  Parameters and ``state_dict`` are untouched.  Returns the number of attention modules patched."""
    def takes(m, names):   # the replacement forwards mirror these call signatures exactly; anything else (a head_mask, ...) stays stock
        try:
            return [p for p in inspect.signature(m.forward).parameters] == names
        except (TypeError, ValueError):
            return False

    n = 0
    for m in module.modules():
        if (all(hasattr(m, a) for a in ("relative_position_bias_table", "relative_position_index", "query", "key", "value", "dropout",
                                        "num_attention_heads", "attention_head_size", "all_head_size"))
                and takes(m, ["hidden_states", "attention_mask", "output_attentions"])
                and not hasattr(m, "_mmk_stock_forward")):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(_window_self_attention_forward, m)
            n += 1
    for m in module.modules():
        if (all(hasattr(m, a) for a in ("layernorm_before", "layernorm_after", "attention", "intermediate", "output", "drop_path", "shift_size",
                                        "window_size", "get_attn_mask", "set_shift_and_window_size"))
                and hasattr(m.attention, "self") and hasattr(m.attention.self, "_mmk_stock_forward") and hasattr(m.attention, "output")
                and takes(m, ["hidden_states", "input_dimensions", "output_attentions", "always_partition"])
                and not hasattr(m, "_mmk_stock_forward")):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(_swin_layer_forward, m)
    return n


class _CubicRowsFn(torch.autograd.Function):
    """``F.interpolate(x, (h_out, w), mode="bicubic", align_corners=True)`` when only the second-to-last axis changes length."""

    @staticmethod
    def forward(ctx, x, h_out):
        ctx.h_in = x.shape[-2]
        return K.cubic_resize_rows(x.contiguous(), h_out)

    @staticmethod
    def backward(ctx, dy):
        return K.cubic_resize_rows(dy.contiguous(), dy.shape[-2], backward_from=ctx.h_in), None


def _reshape_mel2img(self, normalized_input_features):
    """Replaces HF ``ClapAudioEncoder.reshape_mel2img``: the same reshaping, with the bicubic stretch of the time axis (1001 -> 1024 frames)
    by the one-axis kernel of ``csrc/encoder_ops.hip`` instead of ATen's 2-D bicubic (whose four taps along the unchanged mel axis are
    0, 1, 0, 0) -- 1.3 ms each way for 67 MB.  Anything else (a stretch of the mel axis, non-f32 input) goes through the stock method."""
    x = normalized_input_features
    _, _, time_length, freq_length = x.shape
    spec_width = int(self.spec_size * self.freq_ratio)
    spec_height = self.spec_size // self.freq_ratio
    if (not x.is_cuda or x.dtype != torch.float32 or time_length >= spec_width or freq_length != spec_height or freq_length % 4
            or time_length < 2):
        return self._mmk_stock_reshape_mel2img(normalized_input_features)
    x = _CubicRowsFn.apply(x, spec_width)
    batch, channels, time, freq = x.shape
    x = x.reshape(batch, channels * self.freq_ratio, time // self.freq_ratio, freq)
    x = x.permute(0, 1, 3, 2).contiguous()
    return x.reshape(batch, channels, freq * self.freq_ratio, time // self.freq_ratio)


def patch_mel_stretch(module: nn.Module) -> int:
    """Give every HF ``ClapAudioEncoder`` inside ``module`` (recognised by ``reshape_mel2img`` / ``spec_size`` / ``freq_ratio``) the
    one-axis bicubic stretch."""
    n = 0
    for m in module.modules():
        if (hasattr(m, "reshape_mel2img") and hasattr(m, "spec_size") and hasattr(m, "freq_ratio")
                and not hasattr(m, "_mmk_stock_reshape_mel2img")):
            m._mmk_stock_reshape_mel2img = m.reshape_mel2img
            m.reshape_mel2img = types.MethodType(_reshape_mel2img, m)
            n += 1
    return n


def accelerate_encoder(module: nn.Module, low_precision_ln: Iterable[str] = (), fuse_qkv: bool = False, fuse_add_ln: bool = False,
                       cls_only="auto", wgrad_linear: bool = False, window_attention: bool = True) -> dict:
    """Swap ``nn.LayerNorm`` -> :class:`LayerNorm` and quick-GELU activations -> :class:`QuickGELU` inside ``module`` (in place);
    with ``fuse_qkv`` also patch the attention modules (:func:`fuse_qkv_attention`), with ``fuse_add_ln`` the residual
    add + LayerNorm pairs (:func:`fuse_add_layer_norm`).

    ``low_precision_ln``: substrings of qualified module names whose LayerNorm may emit the autocast dtype directly
    (only LayerNorms that feed autocast ``Linear`` layers, e.g. ``("layer_norm1", "layer_norm2", "post_layernorm")``
    for HF CLIP); ``norm1`` / ``norm2`` of timm-style pre-LN blocks (mmlearn's own ViT / predictor), ``layer_norm1`` /
    ``layer_norm2`` of HF ``CLIPEncoderLayer`` and ``layernorm_before`` / ``layernorm_after`` of Swin-style layers (HTSAT) get it
    automatically -- they feed nothing but that block's Linears.
    ``cls_only``: :func:`cls_only_last_layer`.  ``"auto"`` (default): switched on exactly when :func:`reads_only_token0` can prove
    that the tower reads token 0 only (an exact saving: same loss, same gradient for every parameter), left off otherwise;
    ``True``: the caller's own promise -- still refused (``ValueError``) when :func:`reads_only_token0` finds a contradiction
    (``use_all_token_embeddings``, ``output_hidden_states``, a declared ``mmk_reads_only_token0 = False``); ``False``: off.
    The decision and its reason are returned under ``"cls_only"``.
    ``wgrad_linear``: :func:`linear_wgrad` on every ``nn.Linear`` left unpatched (towers without a recognised block structure).
    ``window_attention``: :func:`fuse_window_attention` (on by default; a no-op for towers without windowed attention modules).
    Returns the number of modules swapped per kind.
    """
    swapped = {"layernorm": 0, "quick_gelu": 0, "fused_qkv": fuse_qkv_attention(module) if fuse_qkv else 0, "fused_add_ln": 0}
    low = tuple(low_precision_ln)
    for name, parent in list(module.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{name}.{child_name}" if name else child_name
            if type(child) is nn.LayerNorm and len(child.normalized_shape) == 1 and child.normalized_shape[0] % 4 == 0 \
                    and child.normalized_shape[0] <= 2048:
                # norm1 / norm2 of a timm-style pre-LN block feed nothing but its qkv / fc1 Linear: bf16 out is always safe there
                lowp = any(s in full for s in low) or (child_name in ("norm1", "norm2") and _is_preln_block(parent)) \
                    or (type(parent).__name__ == "CLIPEncoderLayer" and child_name in ("layer_norm1", "layer_norm2")) \
                    or (child_name in ("layernorm_before", "layernorm_after") and hasattr(parent, "shift_size")
                        and hasattr(parent, "attention") and hasattr(parent, "intermediate"))   # Swin-style layer (HTSAT): Linears only
                setattr(parent, child_name, LayerNorm.from_torch(child, lowp))
                swapped["layernorm"] += 1
            elif type(child).__name__ in ("QuickGELUActivation", "QuickGELU") and not isinstance(child, QuickGELU):
                setattr(parent, child_name, QuickGELU())
                swapped["quick_gelu"] += 1
    if fuse_add_ln:  # after the LayerNorm swap, so the patched forwards see the modules' low_precision_out flags
        swapped["fused_add_ln"] = fuse_add_layer_norm(module)
        swapped["patch_conv"] = patch_conv_as_gemm(module)
        swapped["embedding"] = patch_embedding_backward(module)
    swapped["window_attention"] = fuse_window_attention(module) if window_attention else 0   # HTSAT / Swin towers; no such module elsewhere
    if window_attention and swapped["window_attention"]:
        swapped["mel_stretch"] = patch_mel_stretch(module)
    if wgrad_linear:
        swapped["linear_wgrad"] = linear_wgrad(module)
    if cls_only not in (True, False, "auto"):
        raise ValueError(f"cls_only must be True, False or 'auto', got {cls_only!r}")
    if cls_only is not False:
        verdict, why = reads_only_token0(module)
        if cls_only is True and verdict is False:
            raise ValueError(f"accelerate_encoder(cls_only=True) refused: {why}")
        proven = verdict is True
        swapped["cls_only"] = why if (proven or cls_only is True) else f"off: {why}"
        if proven or cls_only is True:     # last: it wraps whatever forward the last layer has by now (fused or stock)
            swapped["cls_only_last_layer"] = cls_only_last_layer(module)
    if any(isinstance(v, int) and not isinstance(v, bool) and v > 0 for v in swapped.values()):
        from .compiled import mark_tower

        mark_tower(module)   # under torch.compile the task hands this encoder to the tracer as one operator (mmlearn_amd/compiled.py)
    return swapped
