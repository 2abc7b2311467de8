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
* :func:`accelerate_encoder` swaps those modules in place inside any encoder (HF CLIP / BERT, mmlearn's own ViT).

There is no CPU path: CPU tensors raise.
"""

from __future__ import annotations

import types
from typing import Iterable, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import kernels as K


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
    """(s, y) = (r + dropout(x), LN(r + dropout(x))) in one kernel each way (csrc/encoder_ops.hip)."""

    @staticmethod
    def forward(ctx, x, r, weight, bias, eps, out_dtype, dropout_p, seed):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        r2 = r.contiguous().view(-1, d)
        w32 = None if weight is None else weight.detach().float().contiguous()
        b32 = None if bias is None else bias.detach().float().contiguous()
        s, y, mean, rstd = K.add_layernorm_fwd(x2, r2, w32, b32, eps, out_dtype, dropout_p, seed)
        ctx.save_for_backward(s, w32, mean, rstd)
        ctx.shape, ctx.x_dtype, ctx.dropout_p, ctx.seed = x.shape, x.dtype, dropout_p, seed
        ctx.wb = (weight is not None and weight.requires_grad, bias is not None and bias.requires_grad,
                  None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        s_out = s.view(x.shape)
        return s_out, y.view(x.shape)

    @staticmethod
    def backward(ctx, gs, gy):
        s, w32, mean, rstd = ctx.saved_tensors
        d = s.shape[-1]
        need_w, need_b, wdt, bdt = ctx.wb
        if gy is None:
            gy = torch.zeros(ctx.shape, dtype=torch.float32, device=s.device)
        ds_in = None if gs is None else gs.contiguous().view(-1, d).float()
        dr, dx, dw, db = K.add_layernorm_bwd(s, gy.contiguous().view(-1, d), ds_in, w32, mean, rstd, ctx.x_dtype, need_w or need_b,
                                             ctx.dropout_p, ctx.seed)
        return (dx.view(ctx.shape), dr.view(ctx.shape), dw.to(wdt) if need_w else None, db.to(bdt) if need_b else None,
                None, None, None, None)


def add_layer_norm(x: torch.Tensor, residual: torch.Tensor, ln: nn.LayerNorm, dropout_p: float = 0.0, seed: Optional[int] = None,
                   low_precision_out: Optional[bool] = None):
    """``s = residual + dropout(x)``, ``y = ln(s)`` fused; returns ``(s, y)``.  ``x`` is the sublayer output (autocast
    dtype or f32), ``residual`` the f32 stream.  ``y`` is f32 like ``F.layer_norm`` under autocast unless
    ``low_precision_out`` (default: the module's own ``low_precision_out`` flag) asks for the autocast dtype."""
    K.require_gpu(x)
    if residual.dtype != torch.float32:
        residual = residual.float()
    if low_precision_out is None:
        low_precision_out = bool(getattr(ln, "low_precision_out", False))
    out_dtype = torch.float32
    if low_precision_out and torch.is_autocast_enabled():
        out_dtype = torch.get_autocast_gpu_dtype()
    if dropout_p > 0.0 and seed is None:
        from .attention import draw_seed

        seed = draw_seed()
    return _AddLayerNormFn.apply(x, residual, ln.weight, ln.bias, ln.eps, out_dtype, float(dropout_p), int(seed or 0))


def _ln_fusable(ln, x: torch.Tensor) -> bool:
    return (isinstance(ln, nn.LayerNorm) and len(ln.normalized_shape) == 1 and x.is_cuda and x.shape[-1] == ln.normalized_shape[0]
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and x.dtype in (torch.float32, torch.bfloat16, torch.float16))


def _clip_layer_forward(self, hidden_states, attention_mask=None, **kwargs):
    """Replaces HF ``CLIPEncoderLayer.forward``.  ``residual + attn`` runs inside ``layer_norm2``'s kernel, and
    ``residual + mlp`` inside the NEXT layer's ``layer_norm1`` kernel: this layer still returns the sum (the hidden state
    HF records), with the already-normalised tensor attached to it for the next layer to pick up."""
    residual = hidden_states
    x = getattr(hidden_states, "_mmk_prenormed", None)
    if x is None:
        x = self.layer_norm1(hidden_states)
    a, _ = self.self_attn(hidden_states=x, attention_mask=attention_mask, **kwargs)
    if _ln_fusable(self.layer_norm2, a):
        h, x2 = add_layer_norm(a, residual, self.layer_norm2)
    else:
        h = residual + a
        x2 = self.layer_norm2(h)
    m = self.mlp(x2)
    nxt = getattr(self, "_mmk_next_ln", None)
    if nxt is not None and _ln_fusable(nxt, m):
        out, y = add_layer_norm(m, h, nxt)
        out._mmk_prenormed = y
        return out
    return h + m


def _bert_output_forward(self, hidden_states, input_tensor):
    """Replaces HF ``BertSelfOutput.forward`` / ``BertOutput.forward``: dropout + residual add + LayerNorm in one kernel."""
    h = self.dense(hidden_states)
    if _ln_fusable(self.LayerNorm, h):
        return add_layer_norm(h, input_tensor, self.LayerNorm, self.dropout.p if self.training else 0.0)[1]
    return self.LayerNorm(self.dropout(h) + input_tensor)


_ADD_LN_FORWARDS = {"CLIPEncoderLayer": _clip_layer_forward, "BertSelfOutput": _bert_output_forward, "BertOutput": _bert_output_forward}


def fuse_add_layer_norm(module: nn.Module) -> int:
    """Patch HF ``CLIPEncoderLayer`` / ``BertSelfOutput`` / ``BertOutput`` inside ``module`` (in place, parameters and
    state_dict untouched) so that each residual add (+ hidden dropout) runs inside the following LayerNorm's kernel."""
    n = 0
    for m in module.modules():
        fwd = _ADD_LN_FORWARDS.get(type(m).__name__)
        if fwd is not None and not hasattr(m, "_mmk_stock_layer_forward"):
            m._mmk_stock_layer_forward = m.forward
            m.forward = types.MethodType(fwd, m)
            n += 1
        if isinstance(m, nn.ModuleList) and len(m) > 1 and all(type(c).__name__ == "CLIPEncoderLayer" for c in m):
            for cur, nxt in zip(list(m)[:-1], list(m)[1:]):   # consecutive pre-LN layers: see _clip_layer_forward
                object.__setattr__(cur, "_mmk_next_ln", nxt.layer_norm1)
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


def _fused_qkv(self, hidden_states: torch.Tensor, names, scale: float, dropout_p: float):
    """One GEMM for the three projections + the packed attention kernel; None when the call is not servable."""
    from .attention import attention_qkvpacked

    q, k, v = (getattr(self, n) for n in names)
    if hidden_states.dim() != 3 or not hidden_states.is_cuda:
        return None
    B, L, E = hidden_states.shape
    if E % 64 or L > 256 or q.weight.shape != (E, E) or not (0.0 <= dropout_p < 1.0):
        return None
    if not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16) and q.weight.dtype != torch.bfloat16:
        return None
    w = torch.cat([q.weight, k.weight, v.weight], 0)
    b = None if q.bias is None else torch.cat([q.bias, k.bias, v.bias], 0)
    qkv = F.linear(hidden_states, w, b)                      # [B, L, 3E] in the autocast dtype
    if qkv.dtype != torch.bfloat16:
        return None
    out = attention_qkvpacked(qkv.view(B, L, 3, E // 64, 64), scale, dropout_p)
    return out.reshape(B, L, E)


def _clip_attention_forward(self, hidden_states, attention_mask=None, **kwargs):
    """Replaces HF ``CLIPAttention.forward`` (modeling_clip.py): same parameters, same return contract."""
    if attention_mask is None and getattr(self, "head_dim", 0) == 64:
        ctx = _fused_qkv(self, hidden_states, ("q_proj", "k_proj", "v_proj"), self.scale, self.dropout if self.training else 0.0)
        if ctx is not None:
            return self.out_proj(ctx), None
    return self._mmk_stock_forward(hidden_states, attention_mask, **kwargs)


def _bert_self_attention_forward(self, hidden_states, attention_mask=None, past_key_values=None, **kwargs):
    """Replaces HF ``BertSelfAttention.forward`` (modeling_bert.py); the output ``dense`` lives in ``BertSelfOutput``."""
    if attention_mask is None and past_key_values is None and getattr(self, "attention_head_size", 0) == 64:
        ctx = _fused_qkv(self, hidden_states, ("query", "key", "value"), self.scaling, self.dropout.p if self.training else 0.0)
        if ctx is not None:
            return ctx, None
    return self._mmk_stock_forward(hidden_states, attention_mask, past_key_values, **kwargs)


_QKV_FORWARDS = {"CLIPAttention": _clip_attention_forward, "BertSelfAttention": _bert_self_attention_forward}


def fuse_qkv_attention(module: nn.Module) -> int:
    """Give every HF ``CLIPAttention`` / ``BertSelfAttention`` inside ``module`` the fused-QKV forward (in place; the
    modules, their parameters and the state_dict stay as they are).  Calls the fused path cannot serve (attention mask,
    KV cache, head size != 64, L > 256, no bf16 autocast) run the stock forward.  Returns the number of modules patched."""
    n = 0
    for m in module.modules():
        fwd = _QKV_FORWARDS.get(type(m).__name__)
        if fwd is not None and not hasattr(m, "_mmk_stock_forward"):
            m._mmk_stock_forward = m.forward
            m.forward = types.MethodType(fwd, m)
            n += 1
    return n


def accelerate_encoder(module: nn.Module, low_precision_ln: Iterable[str] = (), fuse_qkv: bool = False, fuse_add_ln: bool = False) -> dict:
    """Swap ``nn.LayerNorm`` -> :class:`LayerNorm` and quick-GELU activations -> :class:`QuickGELU` inside ``module`` (in place);
    with ``fuse_qkv`` also patch the attention modules (:func:`fuse_qkv_attention`), with ``fuse_add_ln`` the residual
    add + LayerNorm pairs (:func:`fuse_add_layer_norm`).

    ``low_precision_ln``: substrings of qualified module names whose LayerNorm may emit the autocast dtype directly
    (only LayerNorms that feed autocast ``Linear`` layers, e.g. ``("layer_norm1", "layer_norm2", "post_layernorm")``
    for HF CLIP).  Returns the number of modules swapped per kind.
    """
    swapped = {"layernorm": 0, "quick_gelu": 0, "fused_qkv": fuse_qkv_attention(module) if fuse_qkv else 0, "fused_add_ln": 0}
    low = tuple(low_precision_ln)
    for name, parent in list(module.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{name}.{child_name}" if name else child_name
            if type(child) is nn.LayerNorm and len(child.normalized_shape) == 1 and child.normalized_shape[0] % 4 == 0 \
                    and child.normalized_shape[0] <= 2048:
                setattr(parent, child_name, LayerNorm.from_torch(child, any(s in full for s in low)))
                swapped["layernorm"] += 1
            elif type(child).__name__ in ("QuickGELUActivation", "QuickGELU") and not isinstance(child, QuickGELU):
                setattr(parent, child_name, QuickGELU())
                swapped["quick_gelu"] += 1
    if fuse_add_ln:  # after the LayerNorm swap, so the patched forwards see the modules' low_precision_out flags
        swapped["fused_add_ln"] = fuse_add_layer_norm(module)
    return swapped
