"""Short-sequence self-attention on MI355X (second step).

``attention(q, k, v, scale, dropout_p)`` = ``dropout(softmax(scale * q k^T + mask)) v`` for ``[B, H, L, 64]`` bf16 operands with
``L <= 256`` -- the shape of every attention call in the encoders of the headline workload (ViT-B/16 L = 197, mmlearn's own ViT
L = 196 / 169, BERT L = 77 with attention dropout 0.1 and the tokenizer's key-padding mask).  One workgroup keeps a whole
(batch, head) problem on chip (``csrc/attention.hip``); the dropout mask is a counter-based function of a per-call seed
drawn from torch's CPU generator (so ``torch.manual_seed`` reproduces it), regenerated in the backward.  The mask is a per-sample
KEY mask (``key_bias``: the records of :func:`key_bias_of`, one per batch, shared by all layers) and / or the causal triangle.
Registered with HF transformers' ``AttentionInterface`` as ``"mmlearn_hip"`` so that ``CLIPVisionModel`` / ``BertModel`` & co. can
select it per config; calls it cannot serve (a mask that is not provably a key-padding mask, other head dims or dtypes) are
forwarded to SDPA unchanged.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K


def supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attention_mask=None, dropout: float = 0.0, is_causal=False) -> bool:
    """Operand test of the kernels.  ``attention_mask``: None, or a mask :func:`key_mask_view` proves to be a key-padding mask."""
    return (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and q.dim() == 4
            and q.shape == k.shape == v.shape and q.shape[-1] == 64 and q.shape[-2] <= 256
            and (attention_mask is None or key_mask_view(attention_mask, q.shape[0], q.shape[2]) is not None)
            and 0.0 <= float(dropout) < 1.0 and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
            and all(s % 8 == 0 for t in (q, k, v) for s in t.stride()[:3]))


def key_mask_view(mask: torch.Tensor, B: int, L: int):
    """``(view [B, L], additive)`` when the SHAPE AND STRIDES of ``mask`` prove it masks keys only -- the same row for every head and
    every query: ``[B, L]``, ``[B, 1, 1, L]``, or ``[B, 1, Lq, L]`` / ``[B, H, Lq, L]`` expanded along the broadcast dims (stride 0).  A
    materialised ``[B, 1, L, L]`` tensor (what HF's mask helpers hand to the attention modules) could hold anything and is refused
    (None): ``fused.py`` knows where such a tensor came from and keeps the 2-D mask instead.  ``additive``: a floating mask in logit
    units (0 / ``finfo.min``) as opposed to a boolean / integer keep-mask."""
    if not isinstance(mask, torch.Tensor) or not mask.is_cuda or mask.shape[0] != B or mask.shape[-1] != L or mask.stride(-1) != 1 and L > 1:
        return None
    if mask.dim() == 4:
        if not all(mask.shape[d] == 1 or mask.stride(d) == 0 for d in (1, 2)):
            return None
        mask = mask[:, 0, 0]
    elif mask.dim() != 2:
        return None
    if mask.dtype in (torch.bool, torch.uint8, torch.int32, torch.int64):
        return mask, False
    if mask.dtype in (torch.float32, torch.bfloat16):
        return mask, True
    return None


def key_bias_of(mask, B: int, L: int, additive: Optional[bool] = None):
    """Key bias records (``kernels.attn_key_bias``) of a mask :func:`key_mask_view` accepts; None otherwise.  A 2-D floating mask is a
    KEEP mask (HF's ``attention_mask`` of ones and zeros) unless ``additive`` says otherwise; a 4-D floating mask is additive."""
    dim = mask.dim() if isinstance(mask, torch.Tensor) else 0
    kv = key_mask_view(mask, B, L)
    if kv is None:
        return None
    view, add = kv
    if additive is None:
        additive = add and dim == 4
    if view.stride(0) < L:   # expanded over the batch as well
        view = view.contiguous()
    if view.dtype == torch.bfloat16 and not additive:
        view = view.float()
    return K.attn_key_bias(view, additive=bool(additive and view.is_floating_point()))


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, dropout_p, seed, key_bias=None, causal=False):
        out, lse = K.attn_fwd(q, k, v, scale, dropout_p, seed, key_bias, causal)   # out: [B, L, H, 64] contiguous
        ctx.save_for_backward(q, k, v, out, lse, key_bias)
        ctx.scale, ctx.dropout_p, ctx.seed, ctx.causal = scale, dropout_p, seed, causal
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, key_bias = ctx.saved_tensors
        dq, dk, dv = K.attn_bwd(q, k, v, out, lse, dout.contiguous(), ctx.scale, ctx.dropout_p, ctx.seed, key_bias=key_bias, causal=ctx.causal)
        return dq, dk, dv, None, None, None, None, None


class _AttentionPacked(torch.autograd.Function):
    """Same kernels on a packed ``[B, L, 3, H, 64]`` QKV tensor (the output of ONE fused projection); the backward writes
    dq / dk / dv straight into one packed gradient, so autograd sees a single tensor in and out."""

    @staticmethod
    def forward(ctx, qkv, scale, dropout_p, seed, key_bias=None, causal=False):
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))   # [B, H, L, 64] views, row stride 3*H*64
        out, lse = K.attn_fwd(q, k, v, scale, dropout_p, seed, key_bias, causal)
        ctx.save_for_backward(qkv, out, lse, key_bias)
        ctx.scale, ctx.dropout_p, ctx.seed, ctx.causal = scale, dropout_p, seed, causal
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, key_bias = ctx.saved_tensors
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        return (K.attn_bwd(q, k, v, out, lse, dout.contiguous(), ctx.scale, ctx.dropout_p, ctx.seed, packed=True, key_bias=key_bias,
                           causal=ctx.causal), None, None, None, None, None)


def _causal_bias(key_bias, causal: bool, like: torch.Tensor, B: int, L: int):
    """The kernels take the causal triangle only together with a key bias record: an all-attended one when there is no padding."""
    if causal and key_bias is None:
        key_bias = K.attn_key_bias(torch.full((B,), L, dtype=torch.int32, device=like.device), L=L)
    return key_bias


def attention_qkvpacked(qkv: torch.Tensor, scale: Optional[float] = None, dropout_p: float = 0.0, seed: Optional[int] = None,
                        key_bias: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """``qkv``: ``[B, L, 3, H, 64]`` bf16 contiguous (``Linear(E, 3E)(x).view(B, L, 3, H, 64)``) -> ``[B, L, H, 64]``.
    ``key_bias``: :func:`key_bias_of` records of a key-padding mask; ``causal``: also mask key j > query i."""
    K.require_gpu(qkv)
    if not (qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[-1] == 64 and qkv.shape[1] <= 256 and qkv.dtype == torch.bfloat16
            and qkv.is_contiguous() and 0.0 <= dropout_p < 1.0):
        raise ValueError("mmlearn_amd.attention_qkvpacked: need a contiguous bf16 [B, L<=256, 3, H, 64] tensor")
    if dropout_p > 0.0 and seed is None:
        seed = draw_seed()
    key_bias = _causal_bias(key_bias, causal, qkv, qkv.shape[0], qkv.shape[1])
    return _AttentionPacked.apply(qkv, float(scale if scale is not None else 0.125), float(dropout_p), int(seed or 0), key_bias, bool(causal))


def draw_seed() -> int:
    """A 63-bit seed from torch's default CPU generator (no device sync; reproducible under ``torch.manual_seed``)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, dropout_p: float = 0.0,
              seed: Optional[int] = None, key_bias: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """q, k, v: ``[B, H, L, 64]`` bf16 (any strides with a contiguous last dim).  Returns ``[B, L, H, 64]`` contiguous,
    i.e. already in the layout the output projection wants (``.reshape(B, L, H*64)`` is free).  ``dropout_p`` drops
    attention probabilities (training-time semantics of ``F.scaled_dot_product_attention(dropout_p=...)``).  ``key_bias``: the
    records of :func:`key_bias_of` for a key-padding mask (``attn_mask=[B, 1, 1, L]`` of SDPA); ``causal`` = ``is_causal``."""
    K.require_gpu(q)
    if not supported(q, k, v, dropout=dropout_p):
        raise ValueError("mmlearn_amd.attention: need bf16 [B, H, L<=256, 64] operands with contiguous last dim")
    if dropout_p > 0.0 and seed is None:
        seed = draw_seed()
    key_bias = _causal_bias(key_bias, causal, q, q.shape[0], q.shape[2])
    return _Attention.apply(q, k, v, float(scale if scale is not None else q.shape[-1] ** -0.5), float(dropout_p), int(seed or 0), key_bias,
                            bool(causal))


def hf_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0, scaling: Optional[float] = None,
                         is_causal: Optional[bool] = None, **kwargs):
    """transformers ``AttentionInterface`` entry: same contract as ``sdpa_attention_forward`` (returns
    ``(attn_output [B, L, H, dh], None)``).  Unsupported calls fall through to SDPA."""
    if supported(query, key, value, attention_mask, dropout, bool(is_causal)):
        kb = None if attention_mask is None else key_bias_of(attention_mask, query.shape[0], query.shape[2])
        return attention(query, key, value, scaling, float(dropout), key_bias=kb, causal=bool(is_causal)), None
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, is_causal=is_causal, **kwargs)


def register_hf_attention(name: str = "mmlearn_hip") -> str:
    """Make ``config._attn_implementation = "mmlearn_hip"`` available to HF models."""
    from transformers import AttentionInterface

    AttentionInterface.register(name, hf_attention_forward)
    return name
